#!/bin/bash
# One GPU-box pass: parity tests, the three bench workloads, rocprofv3 kernel stats of the default bench
# command, and the two PMC passes (FETCH_SIZE / WRITE_SIZE each in its own run).  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-round}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  tail -3 $O/pytest.log
fi
timeout 600 python bench.py > $O/bench_static.json 2> $O/bench_static.err; tail -1 $O/bench_static.json
if [ "${SKIP_MULTI:-0}" != "1" ]; then
  timeout 600 python bench.py --workload objects4 --warmup 150 --steps 100 --cpu-frames 4 > $O/bench_objects4.json 2> $O/bench_objects4.err; tail -1 $O/bench_objects4.json
  timeout 600 python bench.py --workload objects4-gt --warmup 60 --steps 100 --no-cpu-baseline > $O/bench_objects4gt.json 2> $O/bench_objects4gt.err; tail -1 $O/bench_objects4gt.json
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/prof.log 2>&1
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats.txt 2>&1; head -30 $O/kernel_stats.txt
if [ "${SKIP_PMC:-0}" != "1" ]; then
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/pmc_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/pmc_write.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_fetch icp_reduce > $O/pmc_icp.txt 2>&1
  python $R/tools/pmc_summary.py $O/pmc_write icp_reduce >> $O/pmc_icp.txt 2>&1
  cat $O/pmc_icp.txt
  python $R/tools/pmc_summary.py $O/pmc_fetch > $O/pmc_all_fetch.txt 2>&1
  python $R/tools/pmc_summary.py $O/pmc_write > $O/pmc_all_write.txt 2>&1
  rm -rf $O/pmc_fetch/*counter_collection.csv $O/pmc_write/*counter_collection.csv $O/pmc_fetch/*kernel_trace.csv $O/pmc_write/*kernel_trace.csv
fi
rm -f $O/prof/*kernel_trace.csv
