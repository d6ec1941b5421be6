#!/bin/bash
# Runs on the GPU box (via gpurun): what the driver runs at round end, in its order and form - the GPU suite serially with -x,
# smoke(), the bench line with the driver's flags.  usage: gpurun --timeout 1500 -- 'bash tools/gpu_driver_like.sh [tag]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-driver_like}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 1100 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --timeout 400 --durations=8 ; } > $OUT/gpu_tests.log 2>&1
tail -18 $OUT/gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
{ time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ; } 2>&1 | tail -3
python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["windows_ms"], r["value_median"], r["raster_launch_ms"], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline"]["valu_issue_frac"])
print({k: r[k].get("ms_per_step") for k in ("c5", "c4_single_gpu", "train_step") if k in r}, r.get("other_layout"), r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
PY
