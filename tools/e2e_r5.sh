source tools/gpu_step.sh
step t_cc 900 python -m pytest tests/test_gpu_cc.py tests/test_gpu_solve.py tests/test_gpu_big.py tests/test_round5_hygiene.py tests/test_ueg.py -m gpu -x -q
step b_c2 200 python bench.py --nocc 20 --nvirt 80 --steps 20 --warmup 4 --no-cpu-baseline
step b_c3 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline
