#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4k
mkdir -p $OUT
cd $ROOT
for v in w r; do
  GSX_RASTER2D_BWD=$v CMD="python $ROOT/tools/bench_2dgs.py" bash tools/pmc_sq.sh r4k/pmc_$v raster2d_bwd > /dev/null 2>&1
  echo "== variant $v"; cat $OUT/pmc_$v/sq_counters.txt
done
