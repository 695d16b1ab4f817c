#!/bin/bash
# SQ counter pass (own run, --kernel-trace only) of the default bench command, summarised for kernels matching $2 (default icp_reduce)
#   usage: gpu_pmc_sq.sh <outdir> [kernel-substring] ["ENV=V ..."]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
K=${2:-icp_reduce}
E=${3:-}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
env $E timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 ${BENCH_ARGS:-} > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq $K > $O/pmc_${K}_sq.txt 2>&1; cat $O/pmc_${K}_sq.txt
python $R/tools/pmc_summary.py $O/pmc_sq > $O/pmc_all_sq.txt 2>&1
rm -rf $O/pmc_sq
