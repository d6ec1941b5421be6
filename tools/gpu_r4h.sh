#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4h
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/repro_flaky_render.py 2>&1 | tail -30 | tee $OUT/repro.txt
for i in 1 2 3; do timeout 300 python -m pytest -m gpu -q -p no:cacheprovider --timeout 280 "tests/test_gpu_variants.py::test_raster3d_bwd_variants_match_the_default[t]" 2>&1 | tail -2; done | tee $OUT/variants.txt
timeout 600 python -m pytest -m gpu -q -n 4 -p no:cacheprovider --timeout 280 tests/test_gpu_2dgs.py tests/test_strategy_reference_golden.py tests/test_gpu_variants.py 2>&1 | tail -6 | tee $OUT/tests.txt
