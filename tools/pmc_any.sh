#!/bin/bash
# PMC pass over an arbitrary python command, printing mean counters per kernel whose name matches a pattern.
# Usage: tools/pmc_any.sh "<counters>" <kernel-name-substring> -- python tools/bench_2dgs.py
PMC=$1; PAT=$2; shift 3
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_any
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d /tmp/pmc_any -o p -- "$@" > /tmp/pmc_any.log 2>&1
python - "$PAT" <<'PY'
import csv, glob, sys
from collections import defaultdict
pat = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob("/tmp/pmc_any/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void gsx::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k + ": " + ", ".join(f"{c}={sum(x)/len(x):.4g}" for c, x in sorted(cs.items())))
PY
