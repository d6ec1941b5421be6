#!/bin/bash
# round 4, third call: variant W builds against each other (kernel-level), then SQ counters of variants T and W
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4c
mkdir -p $OUT
cd $ROOT
{
echo "t default: $(GSX_RASTER3D_BWD=t timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
echo "w default: $(GSX_RASTER3D_BWD=w timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
for lib in wpre weag wboth wv2; do
  echo "w $lib: $(GSX_RASTER3D_BWD=w GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_$lib.so timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
done
} | tee $OUT/ab.txt
for v in t w; do
  GSX_RASTER3D_BWD=$v CMD="python $ROOT/tools/bench_raster.py --reps 5" bash tools/pmc_sq.sh r4c/pmc_$v raster3d_bwd > /dev/null 2>&1
  echo "== variant $v"; cat $OUT/pmc_$v/sq_counters.txt
done
