#!/bin/bash
# Round-2 GPU pass A: parity of the reworked tracking kernels (both Gauss-Newton schedules, several launch shapes), launch-shape
# sweep of the configs[2] bench, the full default bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/r02a.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
T="tests/test_track_gpu.py tests/test_refpin_gpu.py tests/test_facade_gpu.py tests/test_surfel_gpu.py"
echo "== parity, default schedule (2 launches / iteration), 256x1" | tee $O/pytest.log
timeout 600 python -m pytest $T -m gpu -x -q >> $O/pytest.log 2>&1; echo "rc=$?" | tee -a $O/pytest.log; tail -4 $O/pytest.log
echo "== parity, 3-launch schedule" | tee -a $O/pytest.log
CF_GN_MODE=0 timeout 600 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py -m gpu -x -q > $O/pytest_gn0.log 2>&1; echo "rc=$?" | tee -a $O/pytest_gn0.log; tail -4 $O/pytest_gn0.log
echo "== parity, 256x4 and 1024x2" | tee -a $O/pytest.log
CF_ICP_LAUNCH=256,4 timeout 600 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py -m gpu -x -q > $O/pytest_ppt4.log 2>&1; echo "rc=$?" | tee -a $O/pytest_ppt4.log; tail -4 $O/pytest_ppt4.log
CF_ICP_LAUNCH=1024,2 timeout 600 python -m pytest tests/test_track_gpu.py -m gpu -x -q > $O/pytest_1024x2.log 2>&1; echo "rc=$?" | tee -a $O/pytest_1024x2.log; tail -3 $O/pytest_1024x2.log
echo "== sweep configs[2]"
: > $O/sweep.jsonl
for mode in 1 0; do for shape in "256 1" "256 2" "256 4" "512 4" "1024 1"; do set -- $shape
  timeout 300 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --gn-mode $mode --icp-threads $1 --icp-ppt $2 >> $O/sweep.jsonl 2>> $O/sweep.err
done; done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print("gn", c["gn_mode"], "launch", c["icp_launch"], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"], "B/launch", r["bytes_per_launch"])
PY
echo "== sweep static"
: > $O/sweep_static.jsonl
for mode in 1 0; do for shape in "256 1" "256 4" "1024 1"; do set -- $shape
  timeout 300 python bench.py --workload static --steps 120 --warmup 30 --no-cpu-baseline --no-extras --gn-mode $mode --icp-threads $1 --icp-ppt $2 >> $O/sweep_static.jsonl 2>> $O/sweep.err
done; done
python - <<PY
import json
for l in open("$O/sweep_static.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print("gn", c["gn_mode"], "launch", c["icp_launch"], "fps", d["value"], "ms", d["ms_per_step"], "icp us", r["avg_us"], "frac", r["frac"])
PY
echo "== full default bench"
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json
echo "== crf-spawned variant"
timeout 300 python bench.py --preroll-masks crf --preroll 200 --no-cpu-baseline --no-extras > $O/bench_crf_spawn.json 2>> $O/sweep.err; tail -1 $O/bench_crf_spawn.json | cut -c1-600
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -40 $O/kernel_stats_objects4.txt
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
