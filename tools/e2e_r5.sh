source tools/gpu_step.sh
step t_gemm 300 python -m pytest tests/test_gpu_gemm_plans.py tests/test_gpu_fuzz.py -m gpu -x -q
step b_c2 200 python bench.py --nocc 20 --nvirt 80 --steps 20 --warmup 3 --no-cpu-baseline
step b_30_120 200 python bench.py --nocc 30 --nvirt 120 --steps 10 --warmup 3 --no-cpu-baseline
step b_stub8 300 python bench.py --stub-collectives --as-rank 0 --of 8 --steps 4 --warmup 2 --no-cpu-baseline
step b_c3 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline
