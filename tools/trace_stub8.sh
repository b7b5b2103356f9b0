#!/bin/bash
# kernel trace of rank 0 of 8 with the collectives stubbed (what a rank computes per iteration at N = 8), ONE iteration in launch
# order: gpurun -- 'bash tools/trace_stub8.sh'
export TMPDIR=/tmp
out=gpurun_out/prof_stub8
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d "$out" -o run --output-format csv -- python3 bench.py --stub-collectives --as-rank 0 --of 8 --steps 4 --warmup 2 --no-cpu-baseline > "$out/run.log" 2>&1
f=$(find "$out" -name 'run_kernel_trace.csv' | head -1)
# (4 timed steps, then 4 eager steps under per-GEMM events: 6 from the end is a timed one)
python3 tools/trace_order.py "$f" 6 t2_layouts > gpurun_out/stub_rank0_of8_dispatch_order.txt
tail -2 gpurun_out/stub_rank0_of8_dispatch_order.txt
