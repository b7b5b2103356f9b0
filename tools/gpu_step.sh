#!/bin/bash
# helper for ad-hoc gpurun calls: `step <name> <timeout> <cmd...>` logs into gpurun_out/<name>.log and stops the whole
# call if a step was killed by its timeout
mkdir -p gpurun_out
step() {
    local name=$1 t=$2; shift 2
    echo "== $name"
    timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2> "gpurun_out/$name.err"
    local rc=$?
    echo "   rc=$rc"; tail -n 2 "gpurun_out/$name.log" | cut -c1-600
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
    return 0
}
