ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/segab3; mkdir -p $OUT; cd $ROOT
for sl in 1024 768 1024 768; do echo "== seg_len $sl"; GSPLAT_AMD_SEG_LEN=$sl timeout 600 python tools/bench_reference_profile.py --big --repeats 20 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['batch'], d['channels'], d['scene_grid'], d['packed'], 'fps', d['fps_fwd'], d['fps_bwd'], 'ms', round(1e3/d['fps_fwd'],4), round(1e3/d['fps_bwd'],4))
"; done 2>&1 | tee $OUT/table.txt
