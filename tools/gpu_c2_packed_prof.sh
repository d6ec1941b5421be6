set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r12c2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o t -- python $ROOT/tools/bench_reference_profile.py --only 0 --repeats 20 --stages > $OUT/c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pk -o t -- python $ROOT/bench.py --lean --packed --steps 10 --warmup 3 > $OUT/pk.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dn -o t -- python $ROOT/bench.py --lean --steps 10 --warmup 3 > $OUT/dn.log 2>&1
cd $ROOT
timeout 200 python tools/tile_costs.py garden 5 > $OUT/tile_costs.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
tail -3 $OUT/c2.log | cut -c1-1500
for d in c2 pk dn; do echo == $d; f=$(find $OUT/$d -name "*kernel_stats.csv"); python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us tot={float(r['TotalDurationNs'])/1e6:8.2f}ms")
P
done
cat $OUT/tile_costs.txt | tail -20
