#!/bin/bash
mkdir -p gpurun_out/r4t; O=gpurun_out/r4t
timeout 400 python tools/gpu_isect_check.py check > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
timeout 300 python tools/gpu_isect_check.py bench c4 > $O/c4.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_segments.py -x -q -n 4 -k "isect or pipeline or segment or c4" -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
tail -3 $O/tests.log; grep -c "^OK" $O/check.txt; grep "FAIL\|ISECT CHECK\|rc=" $O/check.txt | head; grep '"auto"\|legacy"' $O/c4.txt | cut -c1-200
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r4t/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'))
print(b['c4_single_gpu']['ms_per_step'], b['c2_garden'].get('stages_ms'), b['c2_garden'].get('fps_fwd'))
PY
