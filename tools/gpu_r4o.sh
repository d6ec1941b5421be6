#!/bin/bash
# f64 sorting network: rates of the instructions involved, bit-exactness against the integer network / oracle, timings
mkdir -p gpurun_out/r4o; O=gpurun_out/r4o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_rate tools/issue_rate.hip 2>/dev/null && timeout 120 /tmp/issue_rate 39 > $O/issue_rate_f64.json 2>$O/issue_rate.err
timeout 400 python tools/gpu_isect_check.py check bench > $O/check.txt 2>&1; echo "check rc=$?" >> $O/check.txt
timeout 300 python tools/gpu_isect_check.py bench c4 > $O/c4.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k isect -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
tail -3 $O/tests.log; grep -c "^OK" $O/check.txt; grep "FAIL\|ISECT CHECK\|rc=" $O/check.txt | head; grep variant $O/check.txt $O/c4.txt | cut -c1-330
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4o/issue_rate_f64.json').read())
for c in r['cases']: print(c['op'], c['chain'], c['waves_per_simd'], c['instr_per_ns_per_simd_by_wall_clock'])
b=json.loads(open('gpurun_out/r4o/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'), {k:v for k,v in b.get('stage_ms',{}).items() if 'isect' in k})
PY
