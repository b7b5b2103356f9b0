# the collective-hook path on the GPU: its tests, the forced one-rank RCCL run (C2 energy check + C3 timing), the stub N = 8 rank
source tools/gpu_step.sh
step t_hook 600 python -m pytest tests/test_collective_hook.py tests/test_bench_launcher.py -m gpu -x -q
PYMES_FORCE_SHARDED=1 step forced_c2 300 python bench.py --gpus 1 --nocc 20 --nvirt 80 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs
PYMES_FORCE_SHARDED=1 step forced_c3 400 python bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs
step stub8 400 python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs --stub-collectives --as-rank 0 --of 8
