#!/bin/bash
# round 3: workgroup size x chunk count of the Gaussian-major intersection's walk kernels (GSX_FUSED_WG) on c4 / c3 / garden
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for wg in ${WGS:-"" 512x512 512x1024 1024x512 512x256}; do
  echo "== GSX_FUSED_WG=$wg"
  GSX_FUSED_WG=$wg GSX_ISECT_PATH=legacy python tools/gpu_isect_check.py benchone c4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', {k:d[k] for k in d if k.startswith('fused') or k in ('sum_ms','wall_ms')})"
  GSX_FUSED_WG=$wg GSX_ISECT_PATH=legacy python tools/gpu_isect_check.py benchone 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', {k:d[k] for k in d if k.startswith('fused') or k in ('sum_ms','wall_ms')})"
  GSX_FUSED_WG=$wg python tools/bench_reference_profile.py --stages --only 0 --repeats 10 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('garden', d['stages']['fwd_ms'])"
done
