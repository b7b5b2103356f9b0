#!/bin/bash
# Everything profiles/rNN/ holds for one round, re-measured with the library as built (run on the GPU box through gpurun):
#     gpurun --timeout 1150 -- 'bash tools/refresh_profiles.sh 1'     (counter passes of C3 and C2)
#     gpurun --timeout 1150 -- 'bash tools/refresh_profiles.sh 2'     (benches, stubs, other configs, traces)
# Results land in gpurun_out/refresh/; copy what is to be judged into profiles/rNN/.
set -u
out=gpurun_out/refresh
mkdir -p $out
export TMPDIR=/tmp
# (each step's own limit: profile_bench.sh runs one stats pass and five counter passes of up to 600 s each under its own
# per-pass timeouts, so the outer limit only has to outlast their sum — an outer kill mid-pass would leave partial profiles)
step() { local lim=500; case "$1" in *_prof) lim=3700;; esac; echo "== $1" >&2; shift; timeout -k 10 $lim "$@"; rc=$?; echo "   rc=$rc" >&2; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
part=${1:-all}
if [ "$part" = "1" ] || [ "$part" = "all" ]; then
step c3_prof bash tools/profile_bench.sh c3 > $out/prof_c3.txt 2>&1
step c2_prof bash tools/profile_bench.sh c2 --nocc 20 --nvirt 80 > $out/prof_c2.txt 2>&1
rm -rf gpurun_out/prof_c3/stats gpurun_out/prof_c3/pmc_* gpurun_out/prof_c2/stats gpurun_out/prof_c2/pmc_*
fi
if [ "$part" = "1" ]; then echo done; exit 0; fi
step c3_cpu python3 bench.py --steps 10 --warmup 3 > $out/bench_c3_with_cpu_baseline.json 2> $out/bench_c3_with_cpu_baseline.err
step c3_dcsd python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dcsd > $out/bench_c3_dcsd.json 2>/dev/null
step c3_qform env PYMES_LADDER_DRESS=0 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_c3_q_form.json 2>/dev/null
step b30 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --nocc 30 --nvirt 120 > $out/bench_30_120.json 2>/dev/null
for n in 2 4 8; do
    step stub$n python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --stub-collectives --as-rank 0 --of $n > $out/stub_rank0_of$n.json 2>/dev/null
done
step stub8r5 python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --stub-collectives --as-rank 5 --of 8 > $out/stub_rank5_of8.json 2>/dev/null
step stub8ot python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --stub-collectives --as-rank 0 --of 8 --owner-tiles > $out/stub_rank0_of8_owner_tiles.json 2>/dev/null
step configs python3 tools/measure_configs.py --only c2,c4,c5,c5dav > $out/configs_c2_c4_c5.jsonl 2> $out/configs.err
step feast python3 tools/measure_configs.py --only c5feast --skip-cpu > $out/config_c5_feast.jsonl 2> $out/feast.err
step dress python3 tools/probe_dress.py > $out/probe_dress_kernel.txt 2>&1
step eom_stats rocprofv3 --kernel-trace --stats -d $out/eom -o run --output-format csv -- python3 tools/eom_prof_many.py > $out/eom_many.txt 2>&1
f=$(find $out/eom -name 'run_kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/eom_sigma_30_120_k4_kernel_stats.csv
rm -rf $out/eom
step trace rocprofv3 --kernel-trace -d $out/tr -o run --output-format csv -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-other-configs --events timed > $out/trace.log 2>&1
python3 tools/trace_order.py $(find $out/tr -name "run_kernel_trace.csv") > $out/bench_c3_dispatch_order.txt; rm -rf $out/tr
step trace2 rocprofv3 --kernel-trace -d $out/tr2 -o run --output-format csv -- python3 bench.py --nocc 20 --nvirt 80 --steps 6 --warmup 3 --no-cpu-baseline > $out/trace2.log 2>&1
python3 tools/trace_order.py $(find $out/tr2 -name "run_kernel_trace.csv") 8 > $out/bench_c2_dispatch_order.txt; rm -rf $out/tr2
step clock rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $out/pmc -o run --output-format csv -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-other-configs --events timed > $out/clock.log 2>&1
python3 tools/dispatch_clock.py $(find $out/pmc -name "run_counter_collection.csv") 400 > $out/bench_c3_dispatch_clock.txt; rm -rf $out/pmc
# round 6: phase launches — same-box A/B, the level structure of a (20,80) iteration, the boundary / counter / union probe
{
  for i in 1 2; do
    for sz in "20 80 50" "7 50 100" "30 120 20"; do
      echo "phase on : $(timeout -k 10 120 python3 tools/c2_prof.py $sz)"
      echo "phase off: $(PYMES_PHASE=0 timeout -k 10 120 python3 tools/c2_prof.py $sz)"
    done
  done
} > $out/phase_ab.txt 2>&1
PYMES_PHASE_LOG=1 PYMES_NO_GRAPH=1 timeout -k 10 120 python3 tools/c2_prof.py 20 80 1 2> $out/phase_log.txt > /dev/null
awk '/flush/{c++} {l[NR]=$0} END{start=0; n=0; for(i=NR;i>0;i--){ if(l[i] ~ /flush/){n++; if(n==9){start=i; break}} } for(i=start;i<=NR;i++) print l[i]}' $out/phase_log.txt > $out/phase_levels_20_80.txt; rm -f $out/phase_log.txt
step probe_phase bash -c '/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_phase tools/probe_phase.hip && /tmp/probe_phase' > $out/probe_phase_boundary_vs_counters.txt 2>&1
# forced one-rank RCCL runs (the collective table on a real communicator) and the EOM traces
PYMES_FORCE_SHARDED=1 step forced_c2 python3 bench.py --gpus 1 --nocc 20 --nvirt 80 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs > $out/forced_one_rank_rccl_c2.json 2>/dev/null
PYMES_FORCE_SHARDED=1 step forced_c3 python3 bench.py --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline --no-other-configs > $out/forced_one_rank_rccl.json 2>/dev/null
step eom_trace bash tools/trace_eom_many.sh > $out/trace_eom_many.log 2>&1
step gsig_trace bash tools/trace_general_sigma.sh > $out/trace_general_sigma.log 2>&1
step stub_trace bash tools/trace_stub8.sh > $out/trace_stub8.log 2>&1
{
  echo "# tools/probe_graph_edge.hip (in-kernel 100-MHz wall-clock stamps, host kept 2 ms ahead by a spinning kernel; 40 repetitions)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_graph_edge tools/probe_graph_edge.hip && timeout -k 10 120 /tmp/probe_graph_edge
  echo "# tools/host_lead.py: a busy wait of D us in front of the update of every (20,80) pass (first call after the residual graph is launched)"
  timeout -k 10 200 python3 tools/host_lead.py
} > $out/probe_graph_edge_and_host_lead.txt 2>&1
rm -rf gpurun_out/prof_c3/stats gpurun_out/prof_c3/pmc_* gpurun_out/prof_c2/stats gpurun_out/prof_c2/pmc_*
echo done
