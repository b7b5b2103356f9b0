#!/bin/bash
# kernel trace of one graph-replayed CCSD iteration at (nocc, nvirt) = ($1, $2), default (20,80): gpurun -- 'bash tools/trace_c2.sh'
export TMPDIR=/tmp
no=${1:-20}; nv=${2:-80}
out=gpurun_out/prof_trace_${no}_${nv}
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 bench.py --nocc $no --nvirt $nv --steps 6 --warmup 4 --no-cpu-baseline --events separate > "$out/run.log" 2>&1
f=$(find "$out" -name 'run_kernel_trace.csv' | head -1)
python3 tools/trace_order.py "$f" 4 > gpurun_out/dispatch_order_${no}_${nv}.txt
tail -2 gpurun_out/dispatch_order_${no}_${nv}.txt
