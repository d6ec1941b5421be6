#!/bin/bash
# Runs on the GPU box (via gpurun): parity of the kernel variants, then the bench line per variant (lean: timed steps only).
set -u
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 300 python -m pytest ${TESTS:-tests/test_gpu_variants.py tests/test_gpu_training.py} -m gpu -q -x -p no:cacheprovider ; } > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
for lib in ${TEST_LIBS:-}; do
  GSX_VARIANT_LIB=$ROOT/gsplat_amd/csrc/$lib timeout 200 python -m pytest tests/test_gpu_variants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
done
run() { name=$1; shift; env "$@" timeout 120 python bench.py --lean --steps 30 > $OUT/bench_$name.json 2> $OUT/bench_$name.err; python - <<PY
import json
try:
    r = json.load(open("$OUT/bench_$name.json")); print("$name", r["value"], "Mpix/s", r["ms_per_step"], "ms/step  raster fwd / bwd launch", r.get("raster_launch_ms", r["roofline"]["launch_ms"]), "ms")
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/bench_$name.err").read()[-1500:])
PY
}
run default A=1
run r GSX_RASTER3D_BWD=r
# every other build of the library found next to the default one (make SUFFIX=_x EXTRA=-D...)
for lib in $ROOT/gsplat_amd/csrc/libgsplat_amd_*.so; do
  v=$(basename $lib .so); v=${v#libgsplat_amd_}
  [ "$v" = torch ] && continue
  run t_$v GSPLAT_AMD_LIB=$lib
done
