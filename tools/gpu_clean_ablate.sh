#!/bin/bash
# decomposition of clean_kernel by timing ablations (diagnostics build, CF_CLEAN_ABLATE: 1 no staging of the texel patch, 2 no 4x4 window,
# 4 no 3x3 depth window; results of the ablated runs are wrong and discarded): rocprofv3 average of the kernel in the default bench command
#   usage: gpu_clean_ablate.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-cleanabl}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for M in 0 4 2 6 7; do
  CF_LIB_DIR=$R/co_fusion_amd/lib_ablate CF_CLEAN_ABLATE=$M timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 20 > $O/line_$M.json 2> $O/prof_$M.log
  echo "--- CF_CLEAN_ABLATE=$M"; python $R/tools/prof_summary.py $O/prof 2>&1 | grep -E "clean_kernel|associate_kernel|kernel  " | tee -a $O/clean_ablate.txt
  rm -rf $O/prof
done
