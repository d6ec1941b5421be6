#!/bin/bash
# Round 5, call A (runs on the GPU box via gpurun): parity of the new one-wave kernels, then kernel-level A/B.
set -u
TAG=${1:-r5a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 400 python -m pytest tests/test_gpu_variants.py tests/test_gpu_ops.py tests/test_gpu_segments.py tests/test_gpu_sparse.py -m gpu -q -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/tests.log 2>&1
tail -25 $OUT/tests.log
br() { tag=$1; shift; env "$@" timeout 150 python tools/bench_raster.py --tag $tag ${ARGS:-} 2>$OUT/br_$tag.err | tail -1 | tee -a $OUT/bench_raster.jsonl; }
br fwd_w_bwd_w A=1
br fwd_q GSX_RASTER3D_FWD=q
for v in fw4 fw6 fwr0; do br $v GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_$v.so; done
ARGS="--absgrad" br abs_w A=1
ARGS="--absgrad" br abs_r GSX_RASTER3D_BWD=r
ARGS="--channels 32" br c32_chunk A=1
ARGS="--channels 32" br c32_r GSX_BWD_W_WIDE_MIN=0
ARGS="--channels 16" br c16_chunk A=1
ARGS="--channels 16" br c16_r GSX_BWD_W_WIDE_MIN=0
ARGS="--channels 8" br c8_chunk GSX_BWD_W_WIDE_MIN=5
ARGS="--channels 8" br c8_r A=1
timeout 240 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print({k: r.get(k) for k in ("value", "ms_per_step", "value_median", "raster_launch_ms")}, r.get("roofline", {}).get("frac"))
print("stages", r.get("stages_ms") or r.get("stage_ms"))
PY
