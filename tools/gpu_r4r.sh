#!/bin/bash
mkdir -p gpurun_out/r4r; O=gpurun_out/r4r
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_reference_shim.py tests/test_gpu_python_bodies.py -x -q -n 4 -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
tail -5 $O/tests.log
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r4r/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'), b['stage_ms_per_step'])
print(b['c4_single_gpu']['ms_per_step'], b['c5']['ms_per_step'], b['train_step']['ms_per_step'], b['other_layout'])
PY
