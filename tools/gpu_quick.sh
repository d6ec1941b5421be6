#!/bin/bash
# quick A/B on the GPU box: optional test selection, then lean bench lines (default lib + every libgsplat_amd_<v>.so)
# usage: gpurun -- 'TESTS="tests/test_gpu_variants.py" bash tools/gpu_quick.sh <tag>'
set -u
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [ -n "${TESTS:-}" ]; then timeout 600 python -m pytest $TESTS -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log; fi
run() { name=$1; shift; for rep in 1 2; do env "$@" timeout 120 python bench.py --lean --steps 30 > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err; done; python - <<PY
import json
for rep in (1, 2):
    try:
        r = json.load(open("$OUT/bench_${name}_%d.json" % rep)); print("$name", r["ms_per_step"], "ms/step  raster fwd/bwd", r["raster_launch_ms"])
    except Exception as e:
        print("$name FAILED", e); print(open("$OUT/bench_${name}_%d.err" % rep).read()[-800:])
PY
}
run default A=1
for lib in $ROOT/gsplat_amd/csrc/libgsplat_amd_*.so; do
  [ -f "$lib" ] || continue
  v=$(basename $lib .so); v=${v#libgsplat_amd_}
  [ "$v" = torch ] && continue
  run $v GSPLAT_AMD_LIB=$lib
done
