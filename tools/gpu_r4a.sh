#!/bin/bash
# round 4, first call: host facts, the whole GPU suite (4 workers), then the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $ROOT
{ nproc; free -g; rocm-smi --showclocks --showpower 2>/dev/null | head -30; } > $OUT/host.txt 2>&1
{ time timeout 1500 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider --durations=25 ; } > $OUT/gpu_tests.log 2>&1
tail -60 $OUT/gpu_tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["raster_launch_ms"], r["stage_ms_per_step"])
print({k: r[k].get("ms_per_step") for k in ("c5", "c4_single_gpu", "train_step") if k in r}, r.get("other_layout"))
PY
cat $OUT/host.txt
