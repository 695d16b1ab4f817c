#!/bin/bash
# round 4, first GPU call: micro-benchmark of one-XCD grid barriers, the suites touched by the round's changes, bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 60 tools/microbench/xcd_barrier > $O/xcd_barrier.txt 2>&1; echo "xcd_barrier rc=$?"; cat $O/xcd_barrier.txt
timeout 120 tools/microbench/launch_floor > $O/launch_floor.txt 2>&1; tail -4 $O/launch_floor.txt
timeout 700 python -m pytest ${TESTS:-tests/test_track_gpu.py tests/test_icp_gram_gpu.py tests/test_segment_gpu.py tests/test_facade_gpu.py tests/test_group_gpu.py tests/test_refpin_gpu.py} -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest.log
timeout 300 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "frame_loop_fixture" > $O/pytest2.log 2>&1; echo "tests2 rc=$?"; tail -5 $O/pytest2.log
: > $O/sweep.jsonl
timeout 150 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
timeout 150 python bench.py --workload static --steps 120 --warmup 30 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
timeout 150 python bench.py --workload objects8 --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:10], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"], "boxes", r.get("pixels_in_screen_boxes"))
PY
tail -5 $O/sweep.err
