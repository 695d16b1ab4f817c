#!/bin/bash
# PMC passes of the default bench command (separate --pmc runs with --kernel-trace only, as the guide prescribes): FETCH_SIZE,
# WRITE_SIZE, SQ instruction / wait counters; per-kernel summaries for the {ICP || RGB residual} launches and for all kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$c icp_reduce > $O/pmc_icp_$c.txt 2>&1; cat $O/pmc_icp_$c.txt
  python $R/tools/pmc_summary.py $O/pmc_$c > $O/pmc_all_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq icp_reduce > $O/pmc_icp_sq.txt 2>&1; cat $O/pmc_icp_sq.txt
python $R/tools/pmc_summary.py $O/pmc_sq > $O/pmc_all_sq.txt 2>&1
rm -rf $O/pmc_sq
