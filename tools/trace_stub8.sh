#!/bin/bash
# kernel trace of one rank of eight with the collectives stubbed (what a rank computes per iteration at N = 8), launch order
export TMPDIR=/tmp
out=gpurun_out/tr_stub8
rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-other-configs --stub-collectives --as-rank 0 --of 8 > $out/run.log 2>&1
python3 tools/trace_order.py $(find $out -name "run_kernel_trace.csv") > gpurun_out/stub8_dispatch_order.txt
tail -2 gpurun_out/stub8_dispatch_order.txt
