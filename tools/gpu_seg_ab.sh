#!/bin/bash
# Runs on the GPU box: the segment tests, then the c2 garden row (tools/bench_reference_profile.py --only 0) for a list of settings.
# CFGS="<reuse> <seg_len>;..."  (GSPLAT_AMD_SEG_REUSE, GSPLAT_AMD_SEG_LEN). Usage: tools/gpu_seg_ab.sh <tag>
set -u
TAG=${1:-segab}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_segments.py -q -m gpu -x -p no:cacheprovider > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
IFS=';' read -ra CF <<< "${CFGS:-1 768;0 768}"
for rep in 1 2; do
for cfg in "${CF[@]}"; do
  set -- $cfg
  echo "== reuse=$1 seg_len=$2 (run $rep)"
  GSPLAT_AMD_SEG_REUSE=$1 GSPLAT_AMD_SEG_LEN=$2 timeout 300 python tools/bench_reference_profile.py --only ${ONLY:-0} --repeats 40 --stages 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); s = d.get('stages', {})
    print('fps_fwd', d['fps_fwd'], 'fps_bwd', d['fps_bwd'], 'ms', round(1e3/d['fps_fwd'] + 1e3/d['fps_bwd'], 4), 'raster fwd', s.get('fwd_ms', {}).get('raster3d_fwd'), 'bwd', {k: v for k, v in s.get('bwd_ms', {}).items() if 'raster' in k})
"
done
done 2>&1 | tee $OUT/ab.txt
