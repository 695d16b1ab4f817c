#!/bin/bash
out=gpurun_out/${1:-r05b}; mkdir -p $out; : > $out/lines.jsonl
timeout 600 python -m pytest tests/test_icp_gram_gpu.py "tests/test_configs_gpu.py::test_hip_trajectory_within_1mm_ate_of_the_reference_arithmetic" -q -s > $out/pytest_gram.log 2>&1
echo "pytest rc=$?" | tee -a $out/log.txt; grep -E "ATE|passed|failed" $out/pytest_gram.log | tail -8
run() { echo "== $*" >> $out/log.txt; timeout 300 python bench.py "$@" 2>> $out/log.txt | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); d['args'] = '$*'; print(json.dumps(d))
except Exception as e:
    print(json.dumps({'args': '$*', 'error': str(e), 'raw': l[:200]}))" >> $out/lines.jsonl; }
for A in product gram; do
  run --icp-arith $A --no-cpu-baseline --no-extras --workload big --steps 40 --warmup 10
  run --icp-arith $A --no-cpu-baseline --no-extras --workload objects4 --streams 3 --lockstep
done
python - <<PY
import json
for l in open('$out/lines.jsonl'):
    d = json.loads(l); r = d.get('roofline', {})
    print(d.get('args'), '| fps', d.get('value'), '| icp us', r.get('avg_us'), 'frac', r.get('frac'), d.get('error'))
PY
for A in product gram; do
  bash tools/gpu_pmc_sq.sh ${1:-r05b}/sq_$A icp_reduce "CF_ICP_ARITH=$A" | tail -12
done
