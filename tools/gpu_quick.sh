#!/bin/bash
# quick GPU check: a chosen set of tests (TESTS="..." [K="expr"]), configs[2]/[1] bench lines, optional rocprof (PROF=1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-quick}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ -n "${K:-}" ]; then
  timeout ${TEST_TIMEOUT:-600} python -m pytest ${TESTS:-tests/test_facade_gpu.py tests/test_track_gpu.py} -m gpu -x -q -k "$K" > $O/pytest.log 2>&1; echo "tests rc=$?"
else
  timeout ${TEST_TIMEOUT:-600} python -m pytest ${TESTS:-tests/test_facade_gpu.py tests/test_track_gpu.py} -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"
fi
tail -${TAIL:-4} $O/pytest.log
: > $O/sweep.jsonl
for A in ${LINES:-objects4 static}; do
  timeout 150 python bench.py --workload $A --steps 100 --warmup 20 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:10], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"])
PY
if [ "${PROF:-0}" = "1" ]; then
  cd /tmp
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
  python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -${PROF_LINES:-25} $O/kernel_stats_objects4.txt
  find $O/prof -name "*kernel_trace.csv" -delete
fi
