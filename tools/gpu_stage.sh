#!/bin/bash
# per-stage C-ABI times (bench.py stage table) for the default lib and every libgsplat_amd_<v>.so; optional filter
# usage: gpurun -- 'bash tools/gpu_stage.sh isect'
PAT=${1:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
one() { name=$1; shift; env "$@" python bench.py --steps 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.load(sys.stdin); print('$name', r['ms_per_step'], {k:v for k,v in r['stage_ms_per_step'].items() if '$PAT' in k})"; }
one default A=1
for lib in $ROOT/gsplat_amd/csrc/libgsplat_amd_*.so; do
  [ -f "$lib" ] || continue
  v=$(basename $lib .so); v=${v#libgsplat_amd_}
  [ "$v" = torch ] && continue
  one $v GSPLAT_AMD_LIB=$lib
done
