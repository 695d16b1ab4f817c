#!/bin/bash
# Round-2 GPU pass B (tight budget): oracle speed sanity, core parity with durations, host-segmentation A/B, reduced sweep, default bench, rocprof.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02b}
mkdir -p $O
cd $R
export TMPDIR=/tmp
nproc > $O/box.txt; cat /sys/fs/cgroup/cpu.max >> $O/box.txt 2>&1; python -c "import os;print(len(os.sched_getaffinity(0)), os.cpu_count())" >> $O/box.txt
echo "== parity (default: 2 launches / GN iteration, device segmentation)"
timeout 420 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -15 $O/pytest.log
echo "== parity, host segmentation flavour"
CF_SEG_HOST=1 timeout 200 python -m pytest tests/test_facade_gpu.py -m gpu -x -q -k "multi_model or long_free" > $O/pytest_seghost.log 2>&1; echo "rc=$?" | tee -a $O/pytest_seghost.log; tail -4 $O/pytest_seghost.log
echo "== sweep configs[2]"
: > $O/sweep.jsonl
for cfgl in "1 256 1" "1 256 4" "1 512 4" "0 256 1" "0 256 4"; do set -- $cfgl
  timeout 150 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --gn-mode $1 --icp-threads $2 --icp-ppt $3 >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print("gn", c["gn_mode"], "launch", c["icp_launch"], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"], "B/launch", r["bytes_per_launch"])
PY
echo "== static"
timeout 150 python bench.py --workload static --steps 120 --warmup 30 --no-cpu-baseline --no-extras --icp-ppt 1 > $O/static_ppt1.json 2>> $O/sweep.err; cut -c1-700 $O/static_ppt1.json
timeout 150 python bench.py --workload static --steps 120 --warmup 30 --no-cpu-baseline --no-extras --icp-ppt 4 > $O/static_ppt4.json 2>> $O/sweep.err; cut -c1-700 $O/static_ppt4.json
echo "== full default bench"
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json; tail -3 $O/bench_default.err
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -45 $O/kernel_stats_objects4.txt
find $O/prof -name "*kernel_trace.csv" -delete
