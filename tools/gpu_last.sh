#!/bin/bash
# the round's last pass: GPU suite + smoke on the last commit, rocprofv3 durations of the timed level-0 launches, default line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-last}; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_suite.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest_gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> $O/pytest_gpu_suite.log 2>&1; tail -1 $O/pytest_gpu_suite.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 --event-sampling 1 > $O/bench_line_under_rocprof.json 2> $O/prof.log
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -6 $O/kernel_stats_objects4.txt
python $R/tools/timed_launches.py $O/prof "icp_reduce_kernel<2, 4, false>" 1000 61132800 > $O/icp_level0_timed_launches.txt 2>&1
python $R/tools/timed_launches.py $O/prof "icp_reduce_kernel<2, 0, false>" 100 18124800 >> $O/icp_level0_timed_launches.txt 2>&1; cat $O/icp_level0_timed_launches.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats.csv
rm -rf $O/prof
cd $R
timeout 200 python bench.py > $O/bench_default_line.json 2> $O/bench.err; tail -c 300 $O/bench_default_line.json
