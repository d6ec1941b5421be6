#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --lean $*"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
# PMC_SETS: ';'-separated counter sets, one rocprofv3 pass each (default: the four sets below)
IFS=';' read -ra SETS <<< "${PMC_SETS:-FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum;SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY;SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR}"
for pmc in "${SETS[@]}"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $OUT/pmc_$name -o pmc -- python $ROOT/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
tail -3 $OUT/*.log; find $OUT -type f ! -name '*.csv' ! -name '*.log' ! -name '*.txt' -delete; du -sh $OUT; find $OUT -name '*.csv' | head -50
python $ROOT/tools/summarize_prof.py $OUT --json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -80
