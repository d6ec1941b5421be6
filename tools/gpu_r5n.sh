#!/bin/bash
set -u
TAG=${1:-r5n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time GSX_RASTER2D_BWD=m timeout 500 python -m pytest tests/test_gpu_2dgs.py -m gpu -q -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/tests_2d_m.log 2>&1
tail -4 $OUT/tests_2d_m.log; grep -n "^E  \|^FAILED" $OUT/tests_2d_m.log | head -12 | cut -c1-300
for v in r m; do
  env GSX_RASTER2D_BWD=$v timeout 200 python tools/bench_2dgs.py 2> $OUT/b2d_$v.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['stage_ms_per_step'].get('raster2d_bwd_ws'), r['stage_ms_per_step'].get('raster2d_fwd'))" | tee -a $OUT/b2d.txt
done
