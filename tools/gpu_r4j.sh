#!/bin/bash
# round 4: the one-wave-per-tile 2DGS backward: parity, then A/B against the reduction kernel on c5
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4j
mkdir -p $OUT
cd $ROOT
timeout 500 python -m pytest -m gpu -q -n 4 -p no:cacheprovider --timeout 280 tests/test_gpu_2dgs.py 2>&1 | tail -12 | tee $OUT/tests.txt
{
echo "2dgs bwd W: $(timeout 200 python tools/bench_2dgs.py 2>/dev/null | tail -1)"
echo "2dgs bwd R: $(GSX_RASTER2D_BWD=r timeout 200 python tools/bench_2dgs.py 2>/dev/null | tail -1)"
echo "2dgs bwd R launch order: $(GSX_RASTER2D_BWD=r GSX_RASTER3D_BWD_ORDER=0 timeout 200 python tools/bench_2dgs.py 2>/dev/null | tail -1)"
} | tee $OUT/ab.txt
