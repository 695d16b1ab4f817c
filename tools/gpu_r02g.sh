#!/bin/bash
# enqueue-thread A/B: config parity tests, then configs[2] / 8-object benches with 0, 2, 4 helper threads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02g}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_facade_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
: > $O/sweep.jsonl
for T in 0 2 4 7; do
  timeout 150 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --enqueue-threads $T >> $O/sweep.jsonl 2>> $O/sweep.err
done
for T in 0 4 7; do
  timeout 150 python bench.py --workload objects8 --steps 60 --warmup 20 --no-cpu-baseline --no-extras --enqueue-threads $T >> $O/sweep.jsonl 2>> $O/sweep.err
done
timeout 150 python bench.py --workload static --steps 120 --warmup 30 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:10], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"])
PY
tail -5 $O/sweep.err
