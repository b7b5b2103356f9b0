#!/bin/bash
# rocprofv3 evidence for one bench.py workload, run on the GPU box through gpurun:
#     gpurun -- 'bash tools/profile_bench.sh c3'                       (default workload, (50,200))
#     gpurun -- 'bash tools/profile_bench.sh c2 --nocc 20 --nvirt 80'
# Pass 1: --kernel-trace --stats of the bench command itself (per-kernel time, to be compared with the HIP-event
# figures of the bench line written next to it).  Passes 2..6: one --pmc counter each, in runs of their own with
# --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE for HBM traffic, SQ_VALU_MFMA_BUSY_CYCLES /
# GRBM_GUI_ACTIVE / SQ_INSTS_MFMA for MFMA utilisation).  Summaries land in gpurun_out/prof_<tag>/summary/; copy the
# ones to be judged into profiles/rNN/.  The program after `--` is python3 itself (no env / bash -c hop).
set -u
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p "$out/summary"
export TMPDIR=/tmp
hash=$(sha256sum pymes_amd/csrc/kernels.hip | cut -d' ' -f1)
echo "# kernels.hip sha256=$hash" > "$out/summary/kernels_hash.txt"
run() {   # name, timeout, command...
    local name=$1 t=$2; shift 2
    echo "== $name"
    timeout -k 10 "$t" "$@" > "$out/$name.log" 2>&1
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi
    return $rc
}
run stats 600 rocprofv3 --kernel-trace --stats -d "$out/stats" -o run --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --events timed "$@" || exit 1
grep '^{' "$out/stats.log" > "$out/summary/bench_${tag}_under_rocprof.json"
f=$(find "$out/stats" -name 'run_kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/summary/bench_${tag}_kernel_stats.csv"
if [ "${PMC:-1}" = "1" ]; then
    for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA; do
        run "pmc_$c" 600 rocprofv3 --kernel-trace --pmc "$c" -d "$out/pmc_$c" -o run --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --events timed "$@" || exit 1
    done
    cc() { find "$out/pmc_$1" -name 'run_counter_collection.csv' | head -1; }
    { echo "# kernels.hip sha256=$hash"; python3 tools/pmc_summary.py "$(cc FETCH_SIZE)" "$(cc WRITE_SIZE)"; } > "$out/summary/bench_${tag}_pmc_hbm_traffic.csv"
    { echo "# kernels.hip sha256=$hash"; python3 tools/mfma_util_summary.py "$(cc SQ_VALU_MFMA_BUSY_CYCLES)" "$(cc GRBM_GUI_ACTIVE)" "$(cc SQ_INSTS_MFMA)"; } > "$out/summary/bench_${tag}_pmc_mfma_util.csv"
fi
# bench.py reads the counter summary from profiles/<round>/ (and accepts it only if its kernels.hip hash is the current one):
# put this run's summary there on the box, so that the bench lines kept below carry `traffic`
round=${ROUND:-r06}
mkdir -p "profiles/$round"
[ -f "$out/summary/bench_${tag}_pmc_hbm_traffic.csv" ] && cp "$out/summary/bench_${tag}_pmc_hbm_traffic.csv" "profiles/$round/"
# the plain (unprofiled) bench line of the same workload, graph replay allowed
run bench 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs "$@" || exit 1
grep '^{' "$out/bench.log" > "$out/summary/bench_${tag}.json"
ls -la "$out/summary"
