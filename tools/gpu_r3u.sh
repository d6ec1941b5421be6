#!/bin/bash
# round 3: the reference's profiling configs with stage times, under the switches that decide the intersection / segment paths
cd ${GRAFT_REPO_ROOT:-.}
for env in "" "GSX_ISECT_PATH=legacy" "GSPLAT_AMD_SEG_LEN=0" "GSX_ISECT_PATH=legacy GSPLAT_AMD_SEG_LEN=0"; do
  echo "== $env"
  env $env python tools/bench_reference_profile.py --repeats 10 --stages 2>/dev/null | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['stages']
    print((d['batch'],d['channels'],d['scene_grid'],d['packed']), 'fps', d['fps_fwd'], d['fps_bwd'], {k:round(v,3) for k,v in s['fwd_ms'].items()}, {k:round(v,3) for k,v in s['bwd_ms'].items()})"
done
