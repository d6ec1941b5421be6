set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
C=$PWD/gsplat_amd/csrc
one() { echo -n "$1 "; GSPLAT_AMD_LIB=$C/libgsplat_amd$2.so python bench.py --no-extra --no-cpu-baseline --windows 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d.get('stage_ms_per_step'); print(d['ms_per_step'], d.get('windows_ms'), 'fwd', s['raster3d_fwd'], 'bwd', s['raster3d_bwd'], 'proj', s['project_ewa_fwd'], s['project_ewa_bwd'], 'sh', s['sh_fwd'], s['sh_bwd'])"; }
for i in 1 2; do
  one default ""; one bilp _bilp; one biilp _biilp; one bmmc _bmmc; one filp _filp; one fmmc _fmmc
done
for i in 1 2; do
  for v in "" _r2ilp; do echo -n "2dgs$v "; GSPLAT_AMD_LIB=$C/libgsplat_amd$v.so python tools/bench_2dgs.py 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(d['ms_per_step'], {k:v for k,v in s.items() if 'raster' in k or 'project' in k})"; done
done
python -m pytest tests/test_gpu_ops.py tests/test_gpu_pipeline.py tests/test_gpu_2dgs.py tests/test_gpu_eval3d.py tests/test_gpu_ut.py -x -q -n 4 2>&1 | tail -3
