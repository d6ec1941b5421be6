#!/bin/bash
# First gpurun call of the next round (one short call): the checks this round ran out of GPU time for.
#   1. the from-world (eval3d) backward kernel against the pinned reference gradients (tests skipped by default);
#   2. the forward compositing on the packed staging layout (-DGSX_FWD_PACK=1) against the default build:
#      parity (tests/test_gpu_variants.py) and the bench line of each.
# Before calling:  make -C gsplat_amd/csrc SUFFIX=_fwdpack EXTRA=-DGSX_FWD_PACK=1
#   gpurun --timeout 900 -- 'bash tools/gpu_next_round.sh'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/next_round
mkdir -p $OUT
cd $ROOT
GSPLAT_AMD_VALIDATE_EVAL3D_BWD=1 timeout 300 python -m pytest tests/test_gpu_eval3d.py -m gpu -q -p no:cacheprovider > $OUT/eval3d_bwd.log 2>&1
tail -15 $OUT/eval3d_bwd.log
TESTS=tests/test_gpu_variants.py TEST_LIBS=$( [ -f gsplat_amd/csrc/libgsplat_amd_fwdpack.so ] && echo libgsplat_amd_fwdpack.so ) bash tools/gpu_variant_ab.sh next_round_ab
