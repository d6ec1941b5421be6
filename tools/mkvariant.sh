#!/bin/bash
# Build libgsplat_amd_<name>.so that differs from the default build in ONE translation unit (A/B builds without recompiling
# the other 16): tools/mkvariant.sh <name> <unit> <extra hipcc flags...>     e.g.  tools/mkvariant.sh b136 raster3d_bwd -DGSX_BWD_T_BATCH=136
set -e
name=$1; unit=$2; shift 2
cd "$(dirname "$0")/../gsplat_amd/csrc"
make -s >/dev/null
mkdir -p build_$name
cp build/*.o build_$name/
rm -f build_$name/$unit.o
make -s SUFFIX=_$name EXTRA="$*" 2>&1 | grep -i "error" || true
ls -la libgsplat_amd_$name.so | awk '{print $5, $9}'
