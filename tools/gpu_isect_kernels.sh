#!/bin/bash
# Per-kernel times of the binned intersection at c3 (rocprofv3 --kernel-trace --stats around tools/gpu_isect_check.py benchone),
# one line per setting. Runs on the GPU box (via gpurun).
#   tools/gpu_isect_kernels.sh <tag> dbg <GSX_ISECT_DBG values...>     ablation bits (csrc/isect_binned.hip; outputs are wrong when set)
#   tools/gpu_isect_kernels.sh <tag> lib base <variant names...>        variant libraries of tools/mkvariant.sh
TAG=$1; MODE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  unset GSX_ISECT_DBG GSPLAT_AMD_LIB
  if [ $MODE = dbg ]; then export GSX_ISECT_DBG=$v; elif [ $v != base ]; then export GSPLAT_AMD_LIB=$R/gsplat_amd/csrc/libgsplat_amd_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/t$v -o t -- python $R/tools/gpu_isect_check.py benchone > $R/gpurun_out/$TAG/$v.log 2>&1
  echo "$v: $(find $R/gpurun_out/$TAG/t$v -name '*kernel_stats.csv' | head -1 | xargs grep 'gsx::bin_\|gsx::tile_plan' | sed 's/(gsx::BinArgs, unsigned int)/()/' | awk -F, '{n=$1; gsub(/"/,"",n); gsub(/\(.*/,"",n); gsub(/.*::/,"",n); printf "%s %.1f  ", n, $4/1000}')"
done
