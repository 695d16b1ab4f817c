#!/bin/bash
# Gram form of the ICP sums (cf_set_icp_arith 1): parity tests (bit-exact against the oracle's Gram mode, ATE against the
# reference's arithmetic), A/B bench lines product | gram over the configurations, SQ counter passes of both forms.
#   usage: gpu_gram.sh <outdir-under-gpurun_out>          (profiles/r05a_*, r05b_* were made by the two halves of this script)
out=gpurun_out/${1:-gram}; mkdir -p $out; : > $out/lines.jsonl
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
  run --icp-arith $A --no-cpu-baseline
  run --icp-arith $A --no-cpu-baseline --no-extras --workload static
  run --icp-arith $A --no-cpu-baseline --no-extras --workload big --steps 40 --warmup 10
  run --icp-arith $A --no-cpu-baseline --no-extras --workload static --streams 12 --lockstep
  run --icp-arith $A --no-cpu-baseline --no-extras --workload objects4 --streams 3 --lockstep
  run --icp-arith $A --no-cpu-baseline --no-extras --icp-ppt 4
done
python - <<PY
import json
for l in open('$out/lines.jsonl'):
    d = json.loads(l); r = d.get('roofline', {})
    print(d.get('args'), '| fps', d.get('value'), '| icp us', r.get('avg_us'), 'frac', r.get('frac'), '| ate', (d.get('ate_m') or {}).get('vs_oracle'), d.get('error'))
PY
for A in product gram; do
  bash tools/gpu_pmc_sq.sh ${1:-gram}/sq_$A icp_reduce "CF_ICP_ARITH=$A" | tail -12
done
