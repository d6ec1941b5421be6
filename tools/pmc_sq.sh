#!/bin/bash
# SQ counter passes over bench.py --lean for the compositing kernels (separate rocprofv3 passes, 8 counters max each).
# usage: gpurun -- '[CMD="python tools/bench_2dgs.py"] bash tools/pmc_sq.sh <tag> [kernel-substring]'
TAG=${1:-pmc}; PAT=${2:-raster3d}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_WAIT_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INST_LEVEL_LDS"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d /tmp/pmc_$i -o p -- ${CMD:-python $ROOT/bench.py --lean --steps 5 --warmup 2} > $OUT/pass_$i.log 2>&1
done
python - "$PAT" > $OUT/sq_counters.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
pat = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void gsx::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, x in sorted(cs.items()):
        print(f"   {c:28s} {sum(x)/len(x):14.5g}  (n={len(x)})")
PY
cat $OUT/sq_counters.txt
