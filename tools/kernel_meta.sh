#!/bin/bash
# Per-kernel register / LDS / scratch usage of one csrc/*.hip file (cross-compiled; no GPU needed).
# Usage: tools/kernel_meta.sh raster3d_bwd [extra hipcc flags]
f=$1; shift
d=$(mktemp -d /tmp/kmeta.XXXX)
cd "$(dirname "$0")/../gsplat_amd/csrc" || exit 1
extra=""; case $f in intersect|projection|projection2d|isect_fused|isect_binned) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $extra "$@" --save-temps=obj -c $f.hip -o $d/$f.o 2>/dev/null
s=$d/$f-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count):" $s | awk '{printf "%s %s  ", $1, $2} /vgpr_spill_count|\.vgpr_count/ {n++} n==2 {print ""; n=0}'
echo; echo "asm: $s"
