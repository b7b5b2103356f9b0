source tools/gpu_step.sh
step t_eom 900 python -m pytest tests/test_eom.py tests/test_feast.py tests/test_gpu_big.py tests/test_round5_hygiene.py -m gpu -x -q
step b_c3 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline
