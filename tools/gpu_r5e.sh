#!/bin/bash
set -u
TAG=${1:-r5e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_segments.py "tests/test_gpu_pipeline.py::test_c3_compositing_gradients_per_element_band" "tests/test_gpu_pipeline.py::test_c2_garden_scene_1080p_matches_oracle" -m gpu -q -n 3 --dist loadfile -rP -p no:cacheprovider ; } > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log; grep -n "^c3 per-element band:" $OUT/tests.log | cut -c1-3400; grep -n "^FAILED\|^E  " $OUT/tests.log | head -10 | cut -c1-600
g() { tag=$1; shift; env "$@" timeout 200 python tools/bench_reference_profile.py --only 0 --stages 2>$OUT/g_$tag.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', r['fps_fwd'], r['fps_bwd'], r['stages']['bwd_ms'])" | tee -a $OUT/garden.txt; }
g w_div4_ord A=1
g w_div2_ord GSX_BWD_SEG_DIV=2
g w_div4_noord GSX_RASTER3D_BWD_ORDER=0
g w_div2_noord GSX_BWD_SEG_DIV=2 GSX_RASTER3D_BWD_ORDER=0
g t GSX_RASTER3D_BWD=t
