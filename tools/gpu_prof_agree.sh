#!/bin/bash
# bench.py's own roofline.avg_us (HIP events on the level-0 tracking launches of the timed steps) beside rocprofv3's kernel trace of the SAME
# run, the tracking launches split by grid (number of models): the two clocks must agree on the 5-model launches.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-agree}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/bench_line_under_rocprof.json 2> $O/prof.log
python $R/tools/prof_summary.py $O/prof icp_reduce > $O/kernel_stats_by_grid.txt 2>&1
rm -rf $O/prof
python - <<PY
import json
d = json.loads(open("$O/bench_line_under_rocprof.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("bench.py (HIP events, timed steps):", r["avg_us"], "us over", r["launches"], "launches; frames/s under the profiler", d["value"])
PY
grep -A12 "by grid" $O/kernel_stats_by_grid.txt
