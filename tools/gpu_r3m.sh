#!/bin/bash
# round 3: per-kernel times of the binned intersection on c3 with the GSX_ISECT_DBG ablations (rocprofv3 kernel trace)
set -u
TAG=${1:-r3m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dbg in ${DBGS:-0 1 3}; do
  rm -rf /tmp/pr_$dbg
  GSX_ISECT_DBG=$dbg timeout 300 rocprofv3 --kernel-trace -d /tmp/pr_$dbg -o t -- python $ROOT/tools/gpu_isect_check.py benchone ${SCENE:-} > $OUT/bench_dbg$dbg.json 2> $OUT/err_dbg$dbg.log
  echo "== dbg $dbg"; tail -1 $OUT/bench_dbg$dbg.json
  python $ROOT/tools/prof_db.py /tmp/pr_$dbg 2>/dev/null | grep -E "bin_|tile_|fused" | tee $OUT/kernels_dbg$dbg.txt
done
