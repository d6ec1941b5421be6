#!/bin/bash
# Round 5, call C: 2DGS backward after the algebra (both kernels), the per-element band report, c5 bench line.
set -u
TAG=${1:-r5c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_2dgs.py tests/test_gpu_variants.py "tests/test_gpu_pipeline.py::test_c3_compositing_gradients_per_element_band" tests/test_gpu_reference_shim.py -m gpu -q -n 3 --dist loadfile -rP -p no:cacheprovider ; } > $OUT/tests.log 2>&1
tail -12 $OUT/tests.log; grep -n "per-element band:" $OUT/tests.log | cut -c1-2400
for v in r w; do
  env GSX_RASTER2D_BWD=$v timeout 200 python tools/bench_2dgs.py > $OUT/b2d_$v.json 2> $OUT/b2d_$v.err; tail -2 $OUT/b2d_$v.json | cut -c1-900
done
