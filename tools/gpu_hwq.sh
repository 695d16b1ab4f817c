cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
for q in 4 8 16; do
  for w in objects4 objects8; do
    GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('GPU_MAX_HW_QUEUES=$q', '$w', 'fps', d['value'], 'ms', d['ms_per_step'], 'icp us', d['roofline']['avg_us'])" | tee -a gpurun_out/r4f/hw_queues.txt
  done
done
