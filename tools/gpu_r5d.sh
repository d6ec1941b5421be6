#!/bin/bash
# Round 5, call D: per-element band (conditioning-aware), W backward over segment slices (garden), tile costs.
set -u
TAG=${1:-r5d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_segments.py "tests/test_gpu_pipeline.py::test_c3_compositing_gradients_per_element_band" "tests/test_gpu_pipeline.py::test_c2_garden_scene_1080p_matches_oracle" -m gpu -q -n 3 --dist loadfile -rP -p no:cacheprovider ; } > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log; grep -n "^c3 per-element band:" $OUT/tests.log | cut -c1-3000
python tools/tile_costs.py garden 5 2>/dev/null | tail -1 | tee $OUT/tile_costs_garden5.json
python tools/tile_costs.py c3 2>/dev/null | tail -1 | tee $OUT/tile_costs_c3.json
g() { tag=$1; shift; env "$@" timeout 200 python tools/bench_reference_profile.py --only 0 --stages 2>$OUT/g_$tag.err | tail -1 | cut -c1-1500 | tee $OUT/g_$tag.json; }
g w_div4 A=1
g w_div2 GSX_BWD_SEG_DIV=2
g w_div1 GSX_BWD_SEG_DIV=1
g t GSX_RASTER3D_BWD=t
g w_noseg GSPLAT_AMD_SEG_LEN=0
