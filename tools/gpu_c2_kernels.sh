ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/segprof; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/o -o t -- python $ROOT/tools/bench_reference_profile.py --only 0 --repeats 20 > $OUT/o.log 2>&1
f=$(find $OUT/o -name "*kernel_stats.csv"); python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us tot={float(r['TotalDurationNs'])/1e6:8.2f}ms")
P
find $OUT -name "*kernel_trace.csv" -delete
