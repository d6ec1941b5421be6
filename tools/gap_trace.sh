#!/bin/bash
# where does the GPU idle inside a step? tools/gap_trace.sh [bench args]  (run on the GPU box)
# rocprofv3 kernel trace of a short lean bench run -> idle gaps between consecutive kernels of the last steps, summed by
# (kernel before the gap -> kernel after the gap).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/gp
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --lean "$@" > /tmp/gap_bench_out.txt 2>&1
tail -1 /tmp/gap_bench_out.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
python - <<'PY'
import csv, glob, collections
p = glob.glob("/tmp/gp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("gsx::", "").replace("at::native::", "")
    return n.split("(")[0].split("<")[0][:40]
names = [short(r["Kernel_Name"]) for r in rows]
# steady state: the last 6 occurrences of the backward compositing kernel delimit 5 full steps
marks = [i for i, n in enumerate(names) if n.startswith("raster3d_bwd")]
lo, hi = marks[-6], marks[-1]
steps = 5
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[lo + 1:hi + 1])
span = int(rows[hi]["End_Timestamp"]) - int(rows[lo]["End_Timestamp"])
print(f"steady state over {steps} steps: span {span / steps / 1e3:.1f} us/step, kernels busy {busy / steps / 1e3:.1f} us/step, "
      f"idle {(span - busy) / steps / 1e3:.1f} us/step")
gaps = collections.Counter(); cnt = collections.Counter()
for i in range(lo, hi):
    g = int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])
    if g > 0:
        gaps[(names[i], names[i + 1])] += g; cnt[(names[i], names[i + 1])] += 1
for (a, b), g in gaps.most_common(14):
    print(f"{g / steps / 1e3:8.1f} us/step  n={cnt[(a, b)] / steps:4.1f}   {a:40s} -> {b}")
PY
