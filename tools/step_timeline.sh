#!/bin/bash
# kernel timeline of ONE steady-state step (between the last two compositing backward launches): start, duration, gap, name
# usage (GPU box): bash tools/step_timeline.sh [anchor-substring] [bench args]     default: c3 dense, anchor raster3d_bwd
ANCHOR=${1:-raster3d_bwd}; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --lean "$@" > /tmp/tl_out.txt 2>&1
python - "$ANCHOR" <<'PY'
import csv, glob, sys
p = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if sys.argv[1] in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"]); prev = t0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("void ", "").replace("gsx::", "").replace("at::native::", "")[:100]
    print("%8.1f %7.1f gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n))
    prev = e
PY
