#!/bin/bash
# Runs on the GPU box (via gpurun): the GPU parity suite, then the default bench line. Usage: tools/gpu_round.sh <tag>
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export GSPLAT_AMD_REFSUITE_OUT=$OUT/reference_suite
{ time timeout ${TEST_TIMEOUT:-420} python -m pytest tests -m gpu -q -n ${TEST_JOBS:-4} --dist loadfile -p no:cacheprovider ; } > $OUT/gpu_tests.log 2>&1
tail -15 $OUT/gpu_tests.log
timeout 240 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; cat $OUT/bench.json
