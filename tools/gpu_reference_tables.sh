set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4n
mkdir -p $OUT
cd $ROOT
timeout 400 python tools/bench_reference_profile.py --big --stages > $OUT/reference_profile_configs.jsonl 2> $OUT/ref.err; tail -7 $OUT/reference_profile_configs.jsonl | cut -c1-240
timeout 200 python tools/bench_batch_scenes.py > $OUT/batch_scenes.jsonl 2> $OUT/batch.err; tail -5 $OUT/batch_scenes.jsonl
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
r = json.load(open("$OUT/bench.json"))
print(r["value"], r["ms_per_step"], r["windows_ms"], r["other_layout"])
PY
