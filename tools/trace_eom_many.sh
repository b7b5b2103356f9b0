#!/bin/bash
# kernel trace of tools/eom_prof_many.py (k stacked sigma vectors at (30,120)) in launch order: gpurun -- 'bash tools/trace_eom_many.sh'
export TMPDIR=/tmp
out=gpurun_out/prof_eomk
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 tools/eom_prof_many.py > "$out/run.log" 2>&1
f=$(find "$out" -name 'run_kernel_trace.csv' | head -1)
python3 tools/trace_last_build.py "$f" > gpurun_out/eom_k4_dispatch_order.txt
cp "$(find "$out" -name 'run_kernel_stats.csv' | head -1)" gpurun_out/eom_k4_kernel_stats.csv
tail -3 "$out/run.log"; tail -3 gpurun_out/eom_k4_dispatch_order.txt
