#!/bin/bash
# verification pass of a build: GPU suite, rocprofv3 kernel trace of the timed launches, counter passes, default line (in that order:
# the traffic JSON the default line quotes is written by the caller from the counter passes, so the line is re-run afterwards)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-final}; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_suite.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest_gpu_suite.log
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 --event-sampling 1 > $O/bench_line_under_rocprof.json 2> $O/prof.log
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -12 $O/kernel_stats_objects4.txt
python $R/tools/timed_launches.py $O/prof "icp_reduce_kernel<2, 4, false>" 1000 61132800 > $O/icp_level0_timed_launches.txt 2>&1; cat $O/icp_level0_timed_launches.txt
python $R/tools/timed_launches.py $O/prof "icp_reduce_kernel<2, 0, false>" 100 18124800 >> $O/icp_level0_timed_launches.txt 2>&1
find $O/prof -name "*kernel_trace.csv" -delete
cd $R
PMC_SQ=${PMC_SQ:-1} bash tools/gpu_pmc.sh ${1:-final}/pmc > $O/pmc.log 2>&1; tail -12 $O/pmc.log
timeout 280 python bench.py > $O/bench_default_line.json 2> $O/bench.err; tail -c 600 $O/bench_default_line.json
