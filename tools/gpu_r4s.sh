#!/bin/bash
mkdir -p gpurun_out/r4s; O=gpurun_out/r4s
timeout 400 python tools/gpu_isect_check.py check bench > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_segments.py -x -q -n 4 -k "isect or pipeline or allocator or segment" -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 300 python bench.py --lean --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
tail -3 $O/tests.log; grep -c "^OK" $O/check.txt; grep "FAIL\|ISECT CHECK\|rc=" $O/check.txt | head; grep '"auto"\|legacy"' $O/check.txt | cut -c1-200
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r4s/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'), {k:v for k,v in b['stage_ms_per_step'].items() if 'isect' in k})
PY
