#!/bin/bash
set -u
TAG=${1:-r5l}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sparse.py -k "raster" -m gpu -q -x -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/tests_m.log 2>&1
tail -4 $OUT/tests_m.log; grep -n "^E  " $OUT/tests_m.log | head -8 | cut -c1-400
br() { tag=$1; ch=$2; shift 2; env "$@" timeout 150 python tools/bench_raster.py --tag $tag --channels $ch --reps 10 2>$OUT/br_$tag.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], 'ch', r['channels'], 'fwd', r['fwd_us_median'], r['fwd_us_min'], 'bwd', r['bwd_us_median'])" | tee -a $OUT/bench_raster.txt; }
for c in 32 16 12; do br m$c $c GSX_RASTER3D_FWD_WIDE_MIN=5; br q$c $c GSX_RASTER3D_FWD_WIDE=q; done
timeout 200 python tools/bench_reference_profile.py --only 3 --stages 2>/dev/null | tail -1 | cut -c1-900 | tee $OUT/garden32_m.json
