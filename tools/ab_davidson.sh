# same-box A/B of the (30,120) Davidson solve: _ab/ build (tools/ab_eom.sh) against the tree's library, alternating
for i in 1 2; do
  PYMES_AMD_LIBRARY=$PWD/_ab/pymes_amd/lib/libpymes_amd.so timeout -k 10 300 python tools/measure_configs.py --only c5dav --skip-cpu | grep -o '"davidson": {"passes": [0-9]*, "wall_s": [0-9.]*\|"hoist_s": [0-9.]*' | tr '\n' ' ' | sed 's/^/old: /'; echo
  timeout -k 10 300 python tools/measure_configs.py --only c5dav --skip-cpu | grep -o '"davidson": {"passes": [0-9]*, "wall_s": [0-9.]*\|"hoist_s": [0-9.]*' | tr '\n' ' ' | sed 's/^/new: /'; echo
done
timeout -k 10 600 python -m pytest tests/test_eom.py tests/test_gpu_big.py -m gpu -x -q -k "eom or sigma or davidson or c5" 2>&1 | tail -3
