#!/bin/bash
# round 4, second call: the tests that failed in r4a + kernel variants, then variant W against variant T (kernel-level A/B)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4b
mkdir -p $OUT
cd $ROOT
{ time timeout 900 python -m pytest -m gpu -q -n 4 -p no:cacheprovider tests/test_gpu_variants.py tests/test_gpu_python_bodies.py \
   tests/test_gpu_reference_shim.py "tests/test_gpu_ops.py::test_golden_rasterize_vs_reference_outputs" \
   "tests/test_gpu_ops.py::test_projection_packed_matches_dense" "tests/test_gpu_2dgs.py::test_projection_2dgs_packed_matches_dense" \
   "tests/test_gpu_pipeline.py::test_rasterization_sparse_grad_layout_and_values" tests/test_gpu_segments.py ; } > $OUT/tests.log 2>&1
tail -25 $OUT/tests.log
for v in t w; do
  for args in "--gaussians 1000000 --channels 3" "--gaussians 1000000 --channels 4" "--gaussians 4000000 --channels 3" "--gaussians 250000 --channels 3 --scale-mult 3"; do
    echo "variant $v $args: $(GSX_RASTER3D_BWD=$v timeout 200 python tools/bench_raster.py --reps 20 $args 2>/dev/null | tail -1)"
  done
done | tee $OUT/ab_w.txt
