#!/bin/bash
# quick per-kernel time table for bench.py on the GPU box: tools/gpu_trace.sh [bench args]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --lean "$@" > /tmp/bench_out.txt 2>&1
tail -1 /tmp/bench_out.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
python - <<PY
import csv,glob
p=glob.glob("/tmp/p/**/*kernel_stats.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(p)),key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step", round(tot/7e6,3))
for r in rows[:${TOPN:-26}]: print(r["Name"].replace("void ","").replace("gsx::","").replace("at::native::","")[:78].ljust(80), str(int(r["Calls"])//7).rjust(3), str(round(float(r["TotalDurationNs"])/7e3,1)).rjust(8))
PY
