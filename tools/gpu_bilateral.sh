#!/bin/bash
# two-pixels-per-lane bilateral filter: parity tests, then A/B (CF_BILATERAL_V1=1: the one-pixel kernel) on the throughput-bound lock-step run
out=gpurun_out/${1:-r05e}; mkdir -p $out; : > $out/lines.jsonl
timeout 400 python -m pytest tests/test_surfel_gpu.py tests/test_facade_gpu.py::test_facade_static_matches_oracle tests/test_facade_gpu.py::test_full_resolution_matches_oracle tests/test_group_gpu.py -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $out/log.txt; tail -3 $out/pytest.log
run() { echo "== $E $*" >> $out/log.txt; env $E timeout 300 python bench.py "$@" 2>> $out/log.txt | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); d['args'] = '$E $*'; print(json.dumps(d))
except Exception as e:
    print(json.dumps({'args': '$E $*', 'error': str(e), 'raw': l[:200]}))" >> $out/lines.jsonl; }
for E in CF_BILATERAL_V1=1 CF_BILATERAL_V1=0; do
  run --no-cpu-baseline --no-extras --workload static --streams 12 --lockstep --groups 3
  run --no-cpu-baseline --no-extras
done
export TMPDIR=/tmp
R=$(pwd); cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras --workload static --steps 60 > /dev/null 2> $R/$out/prof.log
cd $R; python tools/prof_summary.py $out/prof > $out/static_kernel_stats.txt 2>&1; rm -rf $out/prof
grep -E "bilateral" $out/static_kernel_stats.txt | cut -c1-140
python - <<PY
import json
for l in open('$out/lines.jsonl'):
    d = json.loads(l)
    print(d.get('args'), '| fps', d.get('value'), '| ms', d.get('ms_per_step'), d.get('error'))
PY
