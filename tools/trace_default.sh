#!/bin/bash
# the DEFAULT single-rank bench (launch-graph replay, pipelined read-back) under rocprofv3 --kernel-trace: idle gaps of the timed steps
export TMPDIR=/tmp
out=gpurun_out/tr_default
rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-other-configs > $out/run.log 2>&1
tail -c 300 $out/run.log
python3 tools/trace_gaps.py $(find $out -name "run_kernel_trace.csv") 500 > gpurun_out/default_gaps.txt
cat gpurun_out/default_gaps.txt
