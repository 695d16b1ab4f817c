#!/bin/bash
# Round-2 GPU pass D: parity (core + config-size goldens), data-path A/B of the GN loop, rocprof of configs[2].
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02d}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py -m gpu -x -q > $O/pytest_core.log 2>&1; echo "core rc=$?"; tail -3 $O/pytest_core.log
CF_GN_MODE=0 timeout 200 python -m pytest tests/test_track_gpu.py -m gpu -x -q -k "incremental" > $O/pytest_gn0.log 2>&1; echo "gn0 rc=$?"; tail -2 $O/pytest_gn0.log
timeout 500 python -m pytest tests/test_configs_gpu.py -m gpu -x -q --durations=8 > $O/pytest_configs.log 2>&1; echo "configs rc=$?"; tail -14 $O/pytest_configs.log
: > $O/sweep.jsonl
for cfgl in "objects4 1" "objects4 0" "static 1" "static 0"; do set -- $cfgl
  timeout 150 python bench.py --workload $1 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --gn-mode $2 >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:10], "gn", c["gn_mode"], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"])
PY
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -36 $O/kernel_stats_objects4.txt
find $O/prof -name "*kernel_trace.csv" -delete
