#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of the default bench command under a list of environment settings
#   usage: gpu_kernel_ab.sh <outdir> "<kernel-regex>" "<ENV=V ...>" ...      ("-" = no extra environment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-kab}
K=${2:-clean_kernel}
shift 2
mkdir -p $O
export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i+1))
  EE=$E; [ "$E" = "-" ] && EE=""
  cd /tmp
  env $EE timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$i -o p -- python $R/bench.py --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $O/prof$i.log 2>&1
  python $R/tools/prof_summary.py $O/prof$i > $O/kernel_stats_$i.txt 2>&1
  rm -rf $O/prof$i
  echo "== $E" | tee -a $O/summary.txt
  grep -E "$K" $O/kernel_stats_$i.txt | cut -c1-60,73- | tee -a $O/summary.txt
done
