#!/bin/bash
set -u
TAG=${1:-r5j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
CMD="python $ROOT/tools/bench_raster.py --channels 32 --reps 3" bash tools/pmc_sq.sh $TAG/pmc_full raster3d_bwd_m 2>&1 | tail -34
GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_md63.so CMD="python $ROOT/tools/bench_raster.py --channels 32 --reps 3" bash tools/pmc_sq.sh $TAG/pmc_d63 raster3d_bwd_m 2>&1 | tail -34
