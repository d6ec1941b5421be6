#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of the c5 (2DGS) step and of the c4 (4 M x 4 cameras) step. Usage: tools/gpu_c5_c4_kernels.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/c5c4; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o t -- python $ROOT/tools/bench_2dgs.py > $OUT/c5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o t -- python $ROOT/bench.py --workload c4 --lean --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/c4.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
for d in c5 c4; do echo == $d; head -12 $(find $OUT/$d -name "*kernel_stats.csv") | cut -c1-150; done
