#!/bin/bash
# Runs on the GPU box (via gpurun): the whole GPU suite (four workers, per-test timeout so that a hung rendezvous cannot eat the
# budget), the default bench line, then rocprofv3 --kernel-trace --stats + the separate PMC passes for profiles/.
# usage: gpurun --timeout 1800 -- 'GSX_COMMIT=<short sha> bash tools/gpu_round_full.sh [tag]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-round}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 900 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider --timeout 280 --durations=12 ; } > $OUT/gpu_tests.log 2>&1
tail -30 $OUT/gpu_tests.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["windows_ms"], r["instrumented_window_ms"], r["value_median"], r["raster_launch_ms"], r["stage_ms_per_step"])
print({k: r[k].get("ms_per_step") for k in ("c5", "c4_single_gpu", "train_step") if k in r}, r.get("other_layout"), r["c2_garden"].get("ms_fwd_plus_bwd"), r["train_step"].get("l1_only_ms_per_step"), r["c5"]["raster_launch_ms"])
PY
GSX_COMMIT=${GSX_COMMIT:-} bash tools/gpu_profile.sh $TAG > $OUT/profile.log 2>&1
tail -5 $OUT/profile.log
