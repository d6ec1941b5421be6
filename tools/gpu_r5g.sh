#!/bin/bash
set -u
TAG=${1:-r5g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
br() { tag=$1; shift; env "$@" timeout 150 python tools/bench_raster.py --tag $tag --channels 32 --reps 10 2>$OUT/br_$tag.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], 'ch', r['channels'], 'bwd', r['bwd_us_median'], r['bwd_us_min'])" | tee -a $OUT/bench_raster.txt; }
br m A=1
for v in md1 md2 md4 md8 md16 md31 mcv4 mw2; do br $v GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_$v.so; done
CMD="python $ROOT/tools/bench_raster.py --channels 32 --reps 3" bash tools/pmc_sq.sh $TAG/pmc raster3d_bwd_m 2>&1 | tail -45
