# EOM GPU tests + sigma timings (single, k = 4, Davidson, general) through the bench's other_configs leg
source tools/gpu_step.sh
step t_eom 900 python -m pytest tests/test_eom.py tests/test_feast.py tests/test_gpu_big.py -m gpu -x -q -k "eom or sigma or feast or davidson or c5"
step k4 300 python tools/eom_prof_many.py
step cfg 600 python tools/measure_configs.py --only c5,c5dav --skip-cpu
step gsig 200 python tools/probe_general_sigma.py
head -1 gpurun_out/k4.log; cut -c1-300 gpurun_out/cfg.log
