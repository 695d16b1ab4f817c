#!/bin/bash
# kernel statistics (rocprofv3) and bench lines of one workload under a list of environment settings, all on ONE box; a chosen set of tests first
#   usage: gpu_env_kernel_ab.sh <outdir> "<tests>" "<kernel name pattern>" "<ENV=V ...>" "<...>" ...     ("-" = no extra environment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-envab}; mkdir -p $O
T=$2; PAT=$3
shift 3
cd $R
export TMPDIR=/tmp
for E in "$@"; do
  [ "$E" = "-" ] && E=""
  env $E timeout 600 python -m pytest $T -m gpu -x -q > $O/pytest.log 2>&1; echo "[$E] tests rc=$?"; tail -1 $O/pytest.log
done
REPS=${REPS:-3} bash tools/gpu_env_sweep.sh $(basename $O) ${W:-objects4} "$@"
cd /tmp
for E in "$@"; do
  [ "$E" = "-" ] && E=""
  env $E timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 > /dev/null 2> $O/prof.log
  echo "--- [$E]"; python $R/tools/prof_summary.py $O/prof 2>&1 | grep -E "$PAT" | tee -a $O/kernel_stats.txt
  rm -rf $O/prof
done
