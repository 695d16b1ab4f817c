#!/bin/bash
# one rocprofv3 --pmc pass (own run, --kernel-trace only) of the default bench command with the given counters, summarised for kernels
# matching $2;  usage: gpu_pmc_pass.sh <outdir> <kernel-substring> "<COUNTER ...>" [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
K=${2:-icp_reduce}
C=${3:-SQ_INSTS_VALU}
T=${4:-pass}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 ${BENCH_ARGS:-} > $O/pmc_$T.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_$T "$K" > $O/pmc_$T.txt 2>&1; cat $O/pmc_$T.txt
rm -rf $O/pmc_$T
