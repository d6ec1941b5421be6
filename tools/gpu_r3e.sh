#!/bin/bash
# round 3: binned intersection: parity, timing, ablations (GSX_ISECT_DBG: 1 no sort, 2 no deal, 4 no walk, 8 no entry stores)
set -u
TAG=${1:-r3e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 120 tools/bin/issue_rate 29 > $OUT/issue_rate_masks.json 2>/dev/null; python3 -c "
import json,sys
d=json.load(open("$OUT/issue_rate_masks.json"))
for c in d["cases"]:
    print(c["op"],c["chain"],c["waves_per_simd"],c["counter_ticks_per_instr"])
"
timeout 900 python tools/gpu_isect_check.py check > $OUT/isect_check.log 2>&1; echo "check rc=$?"; grep -v "^OK" $OUT/isect_check.log | tail -20
timeout 600 python tools/gpu_isect_check.py bench > $OUT/isect_bench_c3.jsonl 2> $OUT/isect_bench_c3.err; echo "bench c3 rc=$?"; cat $OUT/isect_bench_c3.jsonl
for dbg in 1 2 3 4 8 12; do
  echo "== GSX_ISECT_DBG=$dbg"; GSX_ISECT_DBG=$dbg timeout 120 python tools/bench_isect.py 2>&1 | tail -1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_isect -o isect -- python $ROOT/tools/bench_isect.py > $OUT/prof_isect.log 2>&1; echo "rocprof rc=$?"
cd $ROOT; python tools/prof_db.py $OUT/prof_isect
