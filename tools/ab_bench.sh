# same-box A/B of bench.py workloads: _ab/ build (tools/ab_eom.sh) against the tree's library, alternating.  usage: ab_bench.sh [bench args]
for i in 1 2; do
  PYMES_AMD_LIBRARY=$PWD/_ab/pymes_amd/lib/libpymes_amd.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs "$@" | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/old: /'
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs "$@" | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/new: /'
done
