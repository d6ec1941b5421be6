#!/bin/bash
# round 4, seventh call: the tests that failed in r4f + the new ones, the training-step record with the fused SSIM kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4g
mkdir -p $OUT
cd $ROOT
{ time timeout 700 python -m pytest -m gpu -q -n 4 -p no:cacheprovider --timeout 280 tests/test_losses.py tests/test_strategy_reference_golden.py \
   tests/test_gpu_python_bodies.py tests/test_gpu_reference_shim.py tests/test_gpu_distributed_multirank.py tests/test_gpu_variants.py \
   "tests/test_gpu_pipeline.py::test_concurrent_threads_and_streams_poll_their_own_counts" tests/test_gpu_2dgs.py ; } > $OUT/tests.log 2>&1
tail -25 $OUT/tests.log
for lam in 0.2 0.0; do echo "train_step ssim_lambda=$lam: $(timeout 300 python tools/train_step_bench.py --steps 60 --ssim-lambda $lam 2>&1 | tail -1 | cut -c1-400)"; done | tee $OUT/train.txt
{
echo "w order on : $(timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
echo "w order off: $(GSX_RASTER3D_BWD_ORDER=0 timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
} | tee $OUT/ab.txt
timeout 300 python bench.py --packed --no-extra --no-cpu-baseline --windows 2 > $OUT/bench_packed.json 2> $OUT/bench_packed.err
python - <<PY
import json
r = json.load(open("$OUT/bench_packed.json"))
print("packed", r["ms_per_step"], r["windows_ms"], r["stage_ms_per_step"], r["other_layout"])
PY
