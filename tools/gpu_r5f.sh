#!/bin/bash
set -u
TAG=${1:-r5f}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_ops.py -k "raster" -m gpu -q -x -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/tests_m.log 2>&1
tail -5 $OUT/tests_m.log; grep -n "^E  " $OUT/tests_m.log | head -8 | cut -c1-400
{ time GSX_RASTER3D_BWD_WIDE=r timeout 500 python -m pytest tests/test_gpu_ops.py -k "channels or one_wave" -m gpu -q -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/tests_r.log 2>&1
tail -3 $OUT/tests_r.log
br() { tag=$1; shift; env "$@" timeout 150 python tools/bench_raster.py --tag $tag ${ARGS:-} 2>$OUT/br_$tag.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], 'ch', r['channels'], 'fwd', r['fwd_us_median'], 'bwd', r['bwd_us_median'], r['bwd_us_min'])" | tee -a $OUT/bench_raster.txt; }
for c in 32 16 8 5; do
  ARGS="--channels $c" br m_c$c A=1
  ARGS="--channels $c" br r_c$c GSX_RASTER3D_BWD_WIDE=r
done
ARGS="--channels 8" br chunk_c8 GSX_RASTER3D_BWD_WIDE=r GSX_BWD_W_WIDE=5,8
timeout 200 python tools/bench_reference_profile.py --only 3 --stages 2>/dev/null | tail -1 | cut -c1-900 | tee $OUT/garden32_m.json
GSX_RASTER3D_BWD_WIDE=r timeout 200 python tools/bench_reference_profile.py --only 3 --stages 2>/dev/null | tail -1 | cut -c1-900 | tee $OUT/garden32_r.json
