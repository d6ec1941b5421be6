#!/bin/bash
# c5 (2DGS) step + stage times for the default lib and every libgsplat_amd_<v>.so next to it; run on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
one() { name=$1; shift; env "$@" timeout 120 python tools/bench_2dgs.py 2>/dev/null | python -c "
import json,sys
r=json.load(sys.stdin); s=r['stage_ms_per_step']; print('$name', r['ms_per_step'], 'raster2d fwd/bwd', s['raster2d_fwd'], s['raster2d_bwd'])"; }
one default A=1
one R GSX_RASTER2D_BWD=r
for lib in $ROOT/gsplat_amd/csrc/libgsplat_amd_*.so; do
  [ -f "$lib" ] || continue
  v=$(basename $lib .so); v=${v#libgsplat_amd_}
  [ "$v" = torch ] && continue
  one $v GSPLAT_AMD_LIB=$lib
done
