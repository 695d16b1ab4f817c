#!/bin/bash
# full GPU suite, then the default bench line and the Gram-arithmetic line
out=gpurun_out/${1:-r05c}; mkdir -p $out; : > $out/lines.jsonl
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_suite.log 2>&1
echo "pytest rc=$?" | tee -a $out/log.txt; tail -4 $out/pytest_gpu_suite.log
run() { echo "== $*" >> $out/log.txt; timeout 300 python bench.py "$@" 2>> $out/log.txt | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); d['args'] = '$*'; print(json.dumps(d))
except Exception as e:
    print(json.dumps({'args': '$*', 'error': str(e), 'raw': l[:200]}))" >> $out/lines.jsonl; }
run
run --icp-arith gram --no-cpu-baseline
run --no-cpu-baseline --no-extras --workload static
run --no-cpu-baseline --no-extras --workload static --streams 12 --lockstep --groups 3
python - <<PY
import json
for l in open('$out/lines.jsonl'):
    d = json.loads(l); r = d.get('roofline', {})
    print(d.get('args'), '| fps', d.get('value'), '| icp us', r.get('avg_us'), 'frac', r.get('frac'), '| ate', (d.get('ate_m') or {}).get('vs_oracle'), '| cpu', (d.get('cpu_baseline') or {}).get('value'), d.get('error'))
PY
