#!/bin/bash
# Kernel-level A/B on the GPU box (via gpurun): tools/bench_raster.py once per variant, one line each.
#   gpurun -- 'bash tools/gpu_ab.sh <tag> "<bench_raster args>" name1 "ENV=.. ENV=.." name2 "GSPLAT_AMD_LIB=<variant lib>" ...'
# Variant libraries come from tools/mkvariant.sh (one translation unit rebuilt with extra -D flags).
set -u
TAG=$1; ARGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  env $envs timeout 150 python tools/bench_raster.py --tag $name $ARGS 2>$OUT/br_$name.err | tail -1 | tee -a $OUT/bench_raster.jsonl
done
