source tools/gpu_step.sh
step t_eom 900 python -m pytest tests/test_eom.py tests/test_feast.py tests/test_gpu_big.py -m gpu -x -q -k "eom or sigma or feast or davidson or c5"
step other 300 python tools/probe_other_configs.py
bash tools/trace_eom_many.sh
