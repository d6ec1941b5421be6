ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/trainprof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $ROOT/tools/train_step_bench.py --steps 60 --split-sh > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt | cut -c1-300
f=$(find $OUT/t -name "*kernel_stats.csv"); python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:34]:
    print(f"{r['Name'][:100]:100s} n={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f}us tot={float(r['TotalDurationNs'])/1e6:8.2f}ms {100*float(r['TotalDurationNs'])/tot:5.1f}%")
P
find $OUT -name "*kernel_trace.csv" -delete
