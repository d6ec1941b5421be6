#!/bin/bash
# round 3: split SH / fp16 tests, forced isect paths, self-launching bench, training-step record, reference Python over the shim
set -u
TAG=${1:-r3j}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k "sh or isect_paths or harmonics" > $OUT/tests_sh_isect.log 2>&1; echo "sh/isect rc=$?"; tail -4 $OUT/tests_sh_isect.log
timeout 900 python -m pytest tests/test_gpu_distributed_multirank.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_multirank.log 2>&1; echo "multirank rc=$?"; tail -4 $OUT/tests_multirank.log
if [ -d $ROOT/.scratch_ref/gsplat ]; then
  GSPLAT_REFERENCE_PATH=$ROOT/.scratch_ref timeout 900 python -m pytest tests/test_gpu_reference_shim.py -m gpu -q -p no:cacheprovider -rA > $OUT/reference_shim.log 2>&1; echo "shim rc=$?"; tail -6 $OUT/reference_shim.log
fi
timeout 600 python tools/train_step_bench.py --steps 100 > $OUT/train_step.json 2> $OUT/train_step.err; echo "train rc=$?"; cat $OUT/train_step.json; tail -3 $OUT/train_step.err
