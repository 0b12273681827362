#!/bin/bash
# VGPRs / SGPRs / scratch / LDS of every kernel in the built objects (kart_amd/csrc/build/*.o): what bounds the occupancy
# usage: tools/kernel_resources.sh [object ...]
set -e
LLVM=/opt/rocm/lib/llvm/bin
ROOT=$(cd "$(dirname "$0")/.." && pwd)
objs=("$@"); [ ${#objs[@]} -eq 0 ] && objs=("$ROOT"/kart_amd/csrc/build/*.o)
tmp=$(mktemp -d)
for o in "${objs[@]}"; do
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$o" 2>/dev/null || continue
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/k.co --unbundle
  echo "== $(basename $o)"
  $LLVM/llvm-readelf --notes $tmp/k.co | awk '
    /\.agpr_count:/ {ag=$2}
    /\.group_segment_fixed_size:/ {lds=$2}
    /\.private_segment_fixed_size:/ {scr=$2}
    /\.sgpr_count:/ {sg=$2}
    /\.symbol:/ {sym=$2; gsub(/\.kd$/,"",sym)}
    /\.vgpr_count:/ {vg=$2}
    /\.vgpr_spill_count:/ {sp=$2}
    /\.wavefront_size:/ {printf "%-100s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %5s lds %6s\n", substr(sym,1,100), vg, ag, sg, sp, scr, lds}'
done
rm -rf $tmp
