#!/bin/bash
# PMC passes (separate --pmc runs with --kernel-trace only, as the guide prescribes):
#   1. calibration: tools/microbench/fetch_calib (known bytes through 4 B/lane and 16 B/lane loads / stores) under FETCH_SIZE and
#      WRITE_SIZE -> the correction factor of each counter for each access width, measured on this box;
#   2. FETCH_SIZE / WRITE_SIZE of the default bench command, per kernel (all kernels) and for the {ICP || RGB residual} launches;
#   3. SQ instruction / wait counters of the same command (PMC_SQ=0 skips).
#   usage: gpu_pmc.sh <outdir> ["ENV=V ..."]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
E=${2:-}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -o p -- $R/tools/microbench/fetch_calib 256 5 > $O/calib_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/calib_$c > $O/pmc_calibration_$c.txt 2>&1; cat $O/pmc_calibration_$c.txt
  rm -rf $O/calib_$c
done
for c in FETCH_SIZE WRITE_SIZE; do
  env $E timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$c icp_reduce > $O/pmc_icp_$c.txt 2>&1; cat $O/pmc_icp_$c.txt
  python $R/tools/pmc_summary.py $O/pmc_$c > $O/pmc_all_$c.txt 2>&1
  rm -rf $O/pmc_$c
done
if [ "${PMC_SQ:-1}" = "1" ]; then
  env $E timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_sq.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_sq icp_reduce > $O/pmc_icp_sq.txt 2>&1; cat $O/pmc_icp_sq.txt
  python $R/tools/pmc_summary.py $O/pmc_sq > $O/pmc_all_sq.txt 2>&1
  rm -rf $O/pmc_sq
fi
