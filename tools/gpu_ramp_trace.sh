#!/bin/bash
# Runs on the GPU box: kernel trace of the first 40 c3 steps from process start; prints, per kernel, its duration in calls 3, 6, 10, 20, 35
# (does the step get faster over its first ~20 calls because every kernel does - the device - or because one of them does?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ramp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $ROOT/tools/step_times.py 40 > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv"); python - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':70s} calls   #3     #6    #10    #20    #35")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 36:
        print(f"{k[:70]:70s} {len(v):4d} " + " ".join(f"{v[i]:6.1f}" for i in (3, 6, 10, 20, 35)))
P
rm -f $f
