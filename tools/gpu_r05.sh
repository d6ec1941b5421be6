#!/bin/bash
# Round-2 GPU call: the WHOLE GPU suite serially (as the driver runs it, but without -x so every failure shows),
# with the eval3d backward validation enabled; then the GSX_FWD_PACK forward A/B; then the default bench line.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_r05.sh <tag>'
set -u
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time GSPLAT_AMD_VALIDATE_EVAL3D_BWD=${VALIDATE_EVAL3D:-1} timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 ; } > $OUT/gpu_tests.log 2>&1
tail -40 $OUT/gpu_tests.log
if [ -n "${AB:-}" ]; then
  for lib in $ROOT/gsplat_amd/csrc/libgsplat_amd_*.so; do
    v=$(basename $lib .so); v=${v#libgsplat_amd_}
    [ "$v" = torch ] && continue
    GSX_VARIANT_LIB=$lib timeout 200 python -m pytest tests/test_gpu_variants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
    for rep in 1 2; do
      GSPLAT_AMD_LIB=$lib timeout 120 python bench.py --lean --steps 30 > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err
      timeout 120 python bench.py --lean --steps 30 > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err
    done
    python - <<PY
import json
for n in ("default_1","${v}_1","default_2","${v}_2"):
    try:
        r=json.load(open("$OUT/bench_%s.json"%n)); print(n, r["ms_per_step"], "ms/step", r["raster_launch_ms"])
    except Exception as e: print(n, "FAILED", e)
PY
  done
fi
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; cat $OUT/bench.json
