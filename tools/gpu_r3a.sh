#!/bin/bash
# round 3, call A: issue-rate microbenchmark, binned-intersection parity + timing, per-kernel trace
set -u
TAG=${1:-r3a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
[ -n "${SKIP_RATE:-}" ] || { timeout 300 tools/bin/issue_rate > $OUT/issue_rate.json 2> $OUT/issue_rate.err; echo "issue_rate rc=$?"; }
timeout 900 python tools/gpu_isect_check.py check > $OUT/isect_check.log 2>&1; echo "check rc=$?"; tail -60 $OUT/isect_check.log
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sparse.py -m gpu -q -x -p no:cacheprovider -k "isect or sort or sparse" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
timeout 600 python tools/gpu_isect_check.py bench > $OUT/isect_bench_c3.jsonl 2> $OUT/isect_bench_c3.err; echo "bench c3 rc=$?"; cat $OUT/isect_bench_c3.jsonl
timeout 600 python tools/gpu_isect_check.py bench c4 > $OUT/isect_bench_c4.jsonl 2> $OUT/isect_bench_c4.err; echo "bench c4 rc=$?"; cat $OUT/isect_bench_c4.jsonl
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_isect -o isect -- python $ROOT/tools/bench_isect.py > $OUT/prof_isect.log 2>&1; echo "rocprof rc=$?"
cd $ROOT
f=$(find $OUT/prof_isect -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
