#!/bin/bash
# round 6 measurement pass on ONE box: rocprofv3 kernel trace of the timed launches, counter passes (FETCH_SIZE / WRITE_SIZE, calibrated) of
# the level-0 {ICP || residual} launch at THREE workloads -- configs[2] (5 trackers), configs[1] (one tracker, from the same run's
# pre-roll) and 1280x960 static --, the launch's decomposition by ablation at the same three (diagnostics build), bench lines of all configs
#   usage: gpu_r6_measure.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r6m}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
# 1. kernel trace + statistics of the default command, durations of the timed level-0 launches
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 --event-sampling 1 > $O/bench_line_under_rocprof.json 2> $O/prof.log
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -8 $O/kernel_stats_objects4.txt
python $R/tools/timed_launches.py $O/prof "icp_reduce_kernel<2, 4, false>" 1000 > $O/icp_level0_timed_launches.txt 2>&1
python $R/tools/timed_launches.py $O/prof "icp_reduce_kernel<2, 0, false>" 100 18124800 >> $O/icp_level0_timed_launches.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats.csv
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb -o p -- python $R/bench.py --workload big-static --no-cpu-baseline --no-extras --steps 60 --warmup 20 --event-sampling 1 > $O/bench_line_big_static_under_rocprof.json 2> $O/profb.log
python $R/tools/timed_launches.py $O/profb "icp_reduce_kernel<2, 0, false>" 600 72499200 >> $O/icp_level0_timed_launches.txt 2>&1
rm -rf $O/profb
cat $O/icp_level0_timed_launches.txt
# 2. counter passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -o p -- $R/tools/microbench/fetch_calib 256 5 > $O/calib_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/calib_$c > $O/pmc_calibration_$c.txt 2>&1; cat $O/pmc_calibration_$c.txt
  rm -rf $O/calib_$c
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmc_$c icp_reduce > $O/pmc_icp_$c.txt 2>&1; cat $O/pmc_icp_$c.txt
  rm -rf $O/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcb_$c -o p -- python $R/bench.py --workload big-static --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmcb_$c.log 2>&1
  python $R/tools/pmc_summary.py $O/pmcb_$c icp_reduce > $O/pmc_icp_big_static_$c.txt 2>&1; cat $O/pmc_icp_big_static_$c.txt
  rm -rf $O/pmcb_$c
done
# 3. decomposition of the launch by ablation (diagnostics build): static, objects4, big-static
cd $R
for W in static objects4 big-static; do
  echo "== $W" >> $O/icp_level0_replay_decomposition.txt
  C=40; [ "$W" = "objects4" ] && C=150
  CF_LIB_DIR=$R/co_fusion_amd/lib_ablate CF_ICP_REPLAY=$C timeout 200 python bench.py --workload $W --no-cpu-baseline --no-extras --steps 40 --warmup 10 2>&1 >/dev/null | grep "icp replay" >> $O/icp_level0_replay_decomposition.txt
done
cat $O/icp_level0_replay_decomposition.txt
# 4. bench lines of all configurations on this box
: > $O/bench_lines_all_configs.jsonl
for W in objects4 static objects8 big big-static; do
  timeout 240 python bench.py --workload $W --no-cpu-baseline --no-extras >> $O/bench_lines_all_configs.jsonl 2>> $O/bench_lines.err
done
python - <<PY
import json
for l in open("$O/bench_lines_all_configs.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]; c = d["config"]
    print(c["workload"][:34], "| fps", d["value"], "ms", d["ms_per_step"], "models", c["active_models"], "| level-0", r["avg_us"], "us", r["bytes_per_launch"], "B frac", r["frac"], "ref-work", r["frac_reference_work"])
PY
