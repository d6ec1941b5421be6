#!/bin/bash
# PMC pass over tools/bench_raster.py for one or more library variants (run on the GPU box).
# Usage: tools/pmc_raster.sh "<counters>" variant_suffix...   (suffix "" = the product library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PMC=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  [ "$v" = "-" ] && v=""
  rm -rf /tmp/pmc_$v
  GSPLAT_AMD_LIB=$ROOT/gsplat_amd/csrc/libgsplat_amd$v.so rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d /tmp/pmc_$v -o p -- python $ROOT/tools/bench_raster.py --reps 5 > /tmp/pmc_$v.log 2>&1
  python - "$v" <<'PY'
import csv, glob, sys
from collections import defaultdict
v = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for p in glob.glob(f"/tmp/pmc_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "raster3d" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void gsx::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(f"[{v or 'product'}] {k}: " + ", ".join(f"{c}={sum(x)/len(x):.4g}" for c, x in sorted(cs.items())))
PY
done
