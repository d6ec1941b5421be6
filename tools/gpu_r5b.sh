#!/bin/bash
# Round 5, call B: the new evidence tests, the full GPU suite, kernel A/B of the forward's select form, bench.
set -u
TAG=${1:-r5b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_variants.py "tests/test_gpu_pipeline.py::test_c3_compositing_gradients_per_element_band" tests/test_gpu_ops.py -k "determin or variants or band or one_wave or channels or absgrad" -m gpu -q -x -s -p no:cacheprovider ; } > $OUT/tests_new.log 2>&1
tail -12 $OUT/tests_new.log; grep -n "per-element band" $OUT/tests_new.log | cut -c1-600
{ time timeout 600 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/gpu_tests.log 2>&1
tail -8 $OUT/gpu_tests.log
br() { tag=$1; shift; env "$@" timeout 150 python tools/bench_raster.py --tag $tag ${ARGS:-} 2>$OUT/br_$tag.err | tail -1 | tee -a $OUT/bench_raster.jsonl; }
br q_at1 A=1
br q_at0 GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_at0.so
br q_at1b A=1
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print({k: r.get(k) for k in ("value", "ms_per_step", "value_median", "raster_launch_ms")}, r.get("roofline", {}).get("frac"))
print("stages", r.get("stage_ms_per_step"))
print("cpu", r.get("cpu_baseline"))
print("c5", r.get("c5")); print("c2", {k: v for k, v in (r.get("c2_garden") or {}).items() if k != "stages_ms"})
PY
