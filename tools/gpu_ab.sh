#!/bin/bash
# A/B on ONE box (boxes differ by +-10 %): parity tests, then bench lines with and without an environment switch
#   usage: gpu_ab.sh <outdir> <ENVVAR> [tests...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ab}
V=${2:-CF_NO_PREP_OVERLAP}
shift 2
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout ${TEST_TIMEOUT:-700} python -m pytest ${@:-tests/test_configs_gpu.py tests/test_facade_gpu.py tests/test_track_gpu.py} -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
: > $O/sweep.jsonl
for rep in $(seq 1 ${REPS:-2}); do
  for W in objects4 static; do
    timeout 150 python bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
    env $V=1 timeout 150 python bench.py --workload $W --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
  done
done
python - <<PY
import json
for i, l in enumerate(open("$O/sweep.jsonl")):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print("new " if i % 2 == 0 else "$V ", c["workload"][:10], "fps", d["value"], "ms", d["ms_per_step"], "icp us", r["avg_us"])
PY
