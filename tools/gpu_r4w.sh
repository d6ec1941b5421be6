#!/bin/bash
mkdir -p gpurun_out/r4w; O=gpurun_out/r4w
timeout 600 python -m pytest tests -x -q -n 4 -m gpu -k "packed or sparse" -p no:cacheprovider --timeout 300 > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
for v in "packed far=6" "packed"; do timeout 200 python tools/step_timeline.py $v > "$O/timeline_$(echo $v | tr ' =' '__').txt" 2>&1; done
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
tail -n 3 $O/tests.log; tail -n 2 $O/timeline_packed_far_6.txt $O/timeline_packed.txt
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r4w/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'))
print(b['other_layout'])
PY
