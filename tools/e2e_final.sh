# forced one-rank RCCL runs + the EOM traces, for profiles/rNN
source tools/gpu_step.sh
PYMES_FORCE_SHARDED=1 step forced_c2 300 python bench.py --gpus 1 --nocc 20 --nvirt 80 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs
PYMES_FORCE_SHARDED=1 step forced_c3 400 python bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs
bash tools/trace_eom_many.sh > gpurun_out/trace_eom_many.log 2>&1
bash tools/trace_general_sigma.sh > gpurun_out/trace_general_sigma.log 2>&1
tail -2 gpurun_out/trace_eom_many.log gpurun_out/trace_general_sigma.log
