cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_obj -o p -- python $R/bench.py --workload objects4 --warmup 150 --steps 60 --no-cpu-baseline > $R/gpurun_out/prof_obj.log 2>&1
python $R/tools/prof_summary.py $R/gpurun_out/prof_obj | head -40
rm -f $R/gpurun_out/prof_obj/*kernel_trace.csv
