#!/bin/bash
# kernel trace of tools/probe_general_sigma.py in launch order: gpurun -- 'bash tools/trace_general_sigma.sh'
export TMPDIR=/tmp
out=gpurun_out/prof_gsig
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 400 python3 tools/probe_general_sigma.py > gpurun_out/general_sigma_time.txt 2>&1 &&
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 tools/probe_general_sigma.py > "$out/run.log" 2>&1
f=$(find "$out" -name 'run_kernel_trace.csv' | head -1)
python3 tools/trace_last_build.py "$f" > gpurun_out/general_sigma_dispatch_order.txt
cat gpurun_out/general_sigma_time.txt; tail -30 gpurun_out/general_sigma_dispatch_order.txt
