#!/bin/bash
# Round-2 GPU pass N: full parity suite, the default bench line, bench lines of all configs, rocprof kernel stats of the default command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02n}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py tests/test_surfel_gpu.py tests/test_refpin_gpu.py tests/test_segment_gpu.py -m gpu -x -q > $O/pytest_core.log 2>&1; echo "core rc=$?"; tail -3 $O/pytest_core.log
timeout 400 python -m pytest tests/test_configs_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_configs.log 2>&1; echo "configs+dist rc=$?"; tail -3 $O/pytest_configs.log
: > $O/sweep.jsonl
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc=$?"; cut -c1-600 $O/bench_default.json
timeout 150 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
timeout 150 python bench.py --workload static --steps 120 --warmup 30 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
timeout 300 python bench.py --workload objects8 --steps 60 --warmup 10 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
timeout 400 python bench.py --workload big --steps 30 --warmup 5 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:10], "gn", c["gn_mode"], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "surfels", c["surfels"][:3], "icp us", r["avg_us"], "frac", r["frac"], "B", r["bytes_per_launch"])
PY
tail -5 $O/sweep.err
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -30 $O/kernel_stats_objects4.txt
find $O/prof -name "*kernel_trace.csv" -delete
