#!/bin/bash
# round 4, fourth call: longest-first tile order for variants T and W (kernel-level A/B), variant parity, then a bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4d
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest -m gpu -q -n 4 -p no:cacheprovider tests/test_gpu_variants.py "tests/test_gpu_pipeline.py::test_c3_matches_oracle" 2>&1 | tail -5
{
for v in t w; do for o in 0 1; do
  echo "variant $v order $o c3 D3: $(GSX_RASTER3D_BWD=$v GSX_RASTER3D_BWD_ORDER=$o timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
done; done
for v in t w; do for o in 0 1; do
  echo "variant $v order $o c3 D4: $(GSX_RASTER3D_BWD=$v GSX_RASTER3D_BWD_ORDER=$o timeout 200 python tools/bench_raster.py --reps 20 --channels 4 2>/dev/null | tail -1)"
done; done
for v in t w; do
  echo "variant $v order 1 big: $(GSX_RASTER3D_BWD=$v timeout 200 python tools/bench_raster.py --reps 20 --gaussians 250000 --scale-mult 3 2>/dev/null | tail -1)"
done
} | tee $OUT/ab.txt
for v in t w; do
GSX_RASTER3D_BWD=$v timeout 400 python bench.py --no-extra --no-cpu-baseline > $OUT/bench_$v.json 2> $OUT/bench_$v.err
python - <<PY
import json
r = json.load(open("$OUT/bench_$v.json"))
print("$v", r["value"], r["ms_per_step"], r["windows_ms"], r["value_median"], r["raster_launch_ms"], r["stage_ms_per_step"], r["gpu_state_under_load"])
PY
done
