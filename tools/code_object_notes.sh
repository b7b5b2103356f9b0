#!/bin/bash
# Register / scratch metadata of every kernel in the gfx950 code object, as shipped (same flags as pymes_amd/csrc/Makefile):
#   bash tools/code_object_notes.sh > profiles/rNN/kernels_code_object_notes.txt
set -e
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -instcombine-max-copied-from-constant-users=100000 --cuda-device-only -c pymes_amd/csrc/kernels.hip -Ipymes_amd/csrc -o "$tmp/k.co" 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$tmp/k.co" --targets=hip-amdgcn-amd-amdhsa--gfx950 --output="$tmp/k.elf"
echo "# kernels.hip sha256=$(sha256sum pymes_amd/csrc/kernels.hip | cut -d' ' -f1)"
echo "# llvm-readelf --notes of the gfx950 code object: kernel, vgpr_count, agpr_count, vgpr_spill_count, private_segment_fixed_size (scratch bytes)"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/k.elf" | grep -E "\.name:|\.vgpr_count|vgpr_spill|private_segment_fixed|agpr_count" | paste - - - - - \
  | sed 's/[[:space:]]\+/ /g' | awk '{n=""; v=""; a=""; s=""; p=""; for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1); if($i==".vgpr_count:")v=$(i+1); if($i==".agpr_count:")a=$(i+1); if($i==".vgpr_spill_count:")s=$(i+1); if($i==".private_segment_fixed_size:")p=$(i+1)} print n, "vgpr="v, "agpr="a, "spill="s, "scratch="p}' | (c++filt 2>/dev/null || cat) | sed 's/(anonymous namespace):://g'
rm -rf "$tmp"
