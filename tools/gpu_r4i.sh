#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4i
mkdir -p $OUT
cd $ROOT
{
for i in 1 2; do
echo "w default  : $(timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
echo "w no-slp   : $(GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_noslp.so timeout 200 python tools/bench_raster.py --reps 30 2>/dev/null | tail -1)"
done
echo "w default D4: $(timeout 200 python tools/bench_raster.py --reps 20 --channels 4 2>/dev/null | tail -1)"
echo "w no-slp  D4: $(GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_noslp.so timeout 200 python tools/bench_raster.py --reps 20 --channels 4 2>/dev/null | tail -1)"
} | tee $OUT/ab.txt
timeout 300 python bench.py --no-extra --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["windows_ms"], r["instrumented_window_ms"], r["value_median"], r["raster_launch_ms"])
PY
