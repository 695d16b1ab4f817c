#!/bin/bash
# lock-step groups on their own streams: S sequences in G groups (bench.py --streams S --lockstep --groups G)
out=gpurun_out/r04d; mkdir -p $out; : > $out/lines.jsonl
run() { echo "== $*" >> $out/log.txt; timeout 150 python bench.py "$@" 2>> $out/log.txt | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); d['args'] = '$*'; print(json.dumps(d))
except Exception as e:
    print(json.dumps({'args': '$*', 'error': str(e), 'raw': l[:200]}))" >> $out/lines.jsonl; }
run --workload static --streams 4 --lockstep --groups 2
run --workload static --streams 8 --lockstep --groups 2
run --workload static --streams 12 --lockstep --groups 2
run --workload static --streams 12 --lockstep --groups 3
run --workload static --streams 16 --lockstep --groups 4
run --workload objects4 --streams 4 --lockstep --groups 2
run --workload objects4 --streams 6 --lockstep --groups 2
python - <<'PY'
import json
for l in open('gpurun_out/r04d/lines.jsonl'):
    d = json.loads(l)
    print(d.get('args'), 'fps', d.get('value'), 'ms/step', d.get('ms_per_step'), 'err', d.get('error'))
PY
tail -5 $out/log.txt
