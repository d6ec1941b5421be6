#!/bin/bash
mkdir -p gpurun_out/r5a; O=gpurun_out/r5a
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_sparse.py -x -q -n 4 -k "isect or packed or pipeline or sparse" -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 300 python tools/bench_reference_profile.py --big --stages > $O/reference_profile_configs.jsonl 2> $O/ref.err
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err
tail -n 3 $O/tests.log
python - <<'PY'
import json
for l in open('gpurun_out/r5a/reference_profile_configs.jsonl'):
    if l.startswith('{'):
        r=json.loads(l); st=r.get('stages',{}).get('fwd_ms',{})
        print(r['batch'], r['channels'], r['scene_grid'], r['packed'], r['fps_fwd'], r['fps_bwd'], {k:v for k,v in st.items() if 'isect' in k})
b=json.loads(open('gpurun_out/r5a/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b.get('windows_ms'))
print(b['other_layout'])
print(b['c2_garden'].get('stages_ms'), b['c2_garden'].get('fps_fwd'), b['c2_garden'].get('fps_bwd'), b['train_step']['ms_per_step'])
PY
