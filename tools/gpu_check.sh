#!/bin/bash
# One gpurun call: GPU test suite, then short bench runs (1 rank; 2 gloo ranks sharing the GPU; forced one-rank RCCL).
# A step that is killed by its timeout stops the script (no further GPU step after a kill).
set -u
mkdir -p gpurun_out
step() {   # name, timeout seconds, command...
    local name=$1 t=$2; shift 2
    echo "== $name" | tee -a gpurun_out/check.log
    timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    local rc=$?
    echo "   rc=$rc" | tee -a gpurun_out/check.log
    tail -n 3 "gpurun_out/$name.log" | tee -a gpurun_out/check.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping" | tee -a gpurun_out/check.log; exit $rc; fi
    return $rc
}
: > gpurun_out/check.log
step pytest_gpu 1000 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"}
step bench_c2_1rank 300 python bench.py --nocc 20 --nvirt 80 --steps 5 --warmup 3 --no-cpu-baseline
step bench_c2_2rank_gloo 300 python bench.py --gpus 2 --backend gloo --nocc 20 --nvirt 80 --steps 5 --warmup 3 --no-cpu-baseline
PYMES_FORCE_SHARDED=1 step bench_c2_forced_rccl 300 python bench.py --gpus 1 --nocc 20 --nvirt 80 --steps 5 --warmup 3 --no-cpu-baseline
step bench_c3 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline
PYMES_FORCE_SHARDED=1 step bench_c3_forced_rccl 400 python bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline
echo done | tee -a gpurun_out/check.log
