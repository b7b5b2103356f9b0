# EOM / FEAST GPU tests + the FEAST operator timing: gpurun -- 'bash tools/e2e_eom.sh'
source tools/gpu_step.sh
step t_eom 900 python -m pytest tests/test_eom.py tests/test_feast.py tests/test_gpu_big.py -m gpu -x -q -k "eom or sigma or feast or davidson or c5"
step gsig 200 python tools/probe_general_sigma.py
step feast 400 python tools/measure_configs.py --only c5feast --skip-cpu
