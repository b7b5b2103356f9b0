# same-box A/B of bench.other_configs (C2 / C4 / C5 rows of the bench line): _ab/ build (tools/ab_eom.sh) against the tree's library, alternating
for i in 1 2 3; do
  for w in old new; do
    if [ $w = old ]; then export PYMES_AMD_LIBRARY=$PWD/_ab/pymes_amd/lib/libpymes_amd.so; else unset PYMES_AMD_LIBRARY; fi
    echo "$w: $(timeout -k 10 300 python -c "
import json, bench
o = bench.other_configs(0)
print(json.dumps({k: round(o[k], 4) for k in ('c2_ms', 'c4_ms', 'c5_sigma_ms', 'c5_sigma_k4_ms_per_vector', 'c5_davidson_s')}), o['c2_ok'], o['c4_ok'], o['c5_ok'], o['c5_davidson_ok'])
")"
  done
done
