#!/bin/bash
# round 4, fifth call: variant W as the default: stage-ahead A/B, the whole GPU suite, the bench line, PMC of W
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4e
mkdir -p $OUT
cd $ROOT
{
echo "w stage-ahead: $(timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
echo "w plain stage: $(GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_nostage.so timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
echo "w stage-ahead D4: $(timeout 200 python tools/bench_raster.py --reps 20 --channels 4 2>/dev/null | tail -1)"
echo "w stage-ahead D1: $(timeout 200 python tools/bench_raster.py --reps 20 --channels 1 2>/dev/null | tail -1)"
echo "t D1: $(GSX_RASTER3D_BWD=t timeout 200 python tools/bench_raster.py --reps 20 --channels 1 2>/dev/null | tail -1)"
} | tee $OUT/ab.txt
{ time timeout 1500 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/gpu_tests.log 2>&1
tail -8 $OUT/gpu_tests.log
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["windows_ms"], r["instrumented_window_ms"], r["value_median"], r["raster_launch_ms"], r["stage_ms_per_step"])
print({k: r[k].get("ms_per_step") for k in ("c5", "c4_single_gpu", "train_step") if k in r}, r.get("other_layout"), r["c2_garden"].get("ms_fwd_plus_bwd"), r["train_step"].get("l1_only_ms_per_step"))
print(r["roofline"]["fwd"], r["roofline"]["bwd"], r["roofline"]["fwd_plus_bwd"])
PY
CMD="python $ROOT/tools/bench_raster.py --reps 5" bash tools/pmc_sq.sh r4e/pmc_w raster3d_bwd > /dev/null 2>&1
cat $OUT/pmc_w/sq_counters.txt
