#!/bin/bash
mkdir -p gpurun_out/r4y; O=gpurun_out/r4y
timeout 400 python tools/gpu_isect_check.py check > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py -x -q -n 4 -k "isect or packed" -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
timeout 300 python tools/bench_reference_profile.py --big --stages > $O/reference_profile_configs.jsonl 2> $O/ref.err; tail -n 7 $O/reference_profile_configs.jsonl | cut -c1-260 > $O/ref_tables.log
tail -n 3 $O/tests.log; grep -c "^OK" $O/check.txt; grep "FAIL\|ISECT CHECK\|rc=" $O/check.txt | head
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r4y/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'))
print(b['other_layout'])
print(b['c2_garden'].get('stages_ms'), b['c2_garden'].get('fps_fwd'), b['c2_garden'].get('fps_bwd'))
PY
tail -n 12 $O/ref_tables.log | cut -c1-300
