#!/bin/bash
# Gram form of the ICP sums (cf_set_icp_arith 1): parity tests, then A/B bench lines product | gram
out=gpurun_out/${1:-r05a}; mkdir -p $out; : > $out/lines.jsonl
timeout 600 python -m pytest tests/test_icp_gram_gpu.py "tests/test_configs_gpu.py::test_hip_trajectory_within_1mm_ate_of_the_reference_arithmetic" -x -q -s > $out/pytest_gram.log 2>&1
echo "pytest rc=$?" | tee -a $out/log.txt; tail -5 $out/pytest_gram.log
run() { echo "== $*" >> $out/log.txt; timeout 200 python bench.py "$@" 2>> $out/log.txt | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); d['args'] = '$*'; print(json.dumps(d))
except Exception as e:
    print(json.dumps({'args': '$*', 'error': str(e), 'raw': l[:200]}))" >> $out/lines.jsonl; }
for A in product gram; do
  run --icp-arith $A --no-cpu-baseline
  run --icp-arith $A --no-cpu-baseline --no-extras --workload static
  run --icp-arith $A --no-cpu-baseline --no-extras --workload objects4_1280 --steps 40 --warmup 10
  run --icp-arith $A --no-cpu-baseline --no-extras --workload static --streams 12 --lockstep
  run --icp-arith $A --no-cpu-baseline --no-extras --icp-ppt 4
done
python - <<PY
import json
for l in open('$out/lines.jsonl'):
    d = json.loads(l); r = d.get('roofline', {})
    print(d.get('args'), '| fps', d.get('value'), '| icp us', r.get('avg_us'), 'frac', r.get('frac'), '| ate', (d.get('ate_m') or {}).get('vs_oracle'), d.get('error'))
PY
