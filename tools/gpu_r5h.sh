#!/bin/bash
set -u
TAG=${1:-r5h}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{ time timeout 500 python -m pytest tests/test_gpu_ops.py -k "raster" -m gpu -q -x -n 4 --dist loadfile -p no:cacheprovider ; } > $OUT/tests_m.log 2>&1
tail -4 $OUT/tests_m.log; grep -n "^E  " $OUT/tests_m.log | head -8 | cut -c1-400
br() { tag=$1; ch=$2; shift 2; env "$@" timeout 150 python tools/bench_raster.py --tag $tag --channels $ch --reps 10 2>$OUT/br_$tag.err | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], 'ch', r['channels'], 'bwd', r['bwd_us_median'], r['bwd_us_min'])" | tee -a $OUT/bench_raster.txt; }
br m 32 A=1
for v in md7 md63 mi1 mi4; do br $v 32 GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd_$v.so; done
br m16 16 A=1; br m8 8 A=1; br m5 5 A=1
timeout 200 python tools/bench_reference_profile.py --only 3 --stages 2>/dev/null | tail -1 | cut -c1-900 | tee $OUT/garden32_m.json
