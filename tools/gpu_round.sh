#!/bin/bash
# One verification pass on a GPU box: the whole GPU suite, the default bench line (what the driver runs), bench lines of the other
# configurations, rocprofv3 kernel stats of the default command.  Outputs under gpurun_out/<tag>/.
#   SKIP_TESTS=1 / SKIP_LINES=1 / SKIP_PROF=1 skip a part; LONG=1 adds the COFUSION_LONG_TESTS replays
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-round}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  [ "${LONG:-0}" = "1" ] && export COFUSION_LONG_TESTS=1
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
  tail -4 $O/pytest.log
fi
timeout 600 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; tail -c 600 $O/bench_default_line.json; echo
if [ "${SKIP_LINES:-0}" != "1" ]; then
  : > $O/bench_lines_all_configs.jsonl
  for A in "--workload objects4" "--workload static" "--workload objects8" "--workload big" "--workload big-static" "--workload static --streams 4 --lockstep" "--workload objects4 --streams 3 --lockstep"; do
    timeout 400 python bench.py $A --no-cpu-baseline --no-extras >> $O/bench_lines_all_configs.jsonl 2>> $O/bench_lines.err
  done
  python - <<PY
import json
for l in open("$O/bench_lines_all_configs.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:60].ljust(60), "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"])
PY
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
  python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -24 $O/kernel_stats_objects4.txt
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats.csv
  rm -rf $O/prof
fi
