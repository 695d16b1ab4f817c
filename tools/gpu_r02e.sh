#!/bin/bash
# Round-2 GPU pass E: parity with gather culling, multi-rank paths on one GPU (gloo), culling A/B, PMC passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02e}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_track_gpu.py tests/test_facade_gpu.py tests/test_refpin_gpu.py -m gpu -x -q > $O/pytest_core.log 2>&1; echo "core rc=$?"; tail -3 $O/pytest_core.log
timeout 300 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "objects4_640x480_free_run or nine_models" > $O/pytest_configs.log 2>&1; echo "configs rc=$?"; tail -3 $O/pytest_configs.log
timeout 400 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q > $O/pytest_dist.log 2>&1; echo "dist rc=$?"; tail -12 $O/pytest_dist.log
: > $O/sweep.jsonl
timeout 150 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
CF_NO_CULLING=1 timeout 150 python bench.py --workload objects4 --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:10], "fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "icp us", r["avg_us"], "frac", r["frac"])
PY
echo "== 2 ranks on one GPU (gloo), objects8, models over ranks"
CF_BENCH_BACKEND=gloo CF_BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "rc=$?"; tail -1 $O/bench_2ranks.json | cut -c1-900; tail -5 $O/bench_2ranks.err
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$c icp_reduce > $O/pmc_icp_$c.txt 2>&1; cat $O/pmc_icp_$c.txt
  rm -rf $O/pmc_$c
done
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_sq.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_sq icp_reduce > $O/pmc_icp_sq.txt 2>&1; cat $O/pmc_icp_sq.txt
python $R/tools/pmc_summary.py $O/pmc_sq > $O/pmc_all_sq.txt 2>&1
rm -rf $O/pmc_sq
