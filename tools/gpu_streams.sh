#!/bin/bash
# aggregate throughput of several independent RGB-D streams per GPU (own context + HIP stream + host thread each)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-streams}
mkdir -p $O
cd $R
export TMPDIR=/tmp
: > $O/sweep.jsonl
for W in objects4 static; do
  for S in 1 2 4; do
    timeout 240 python bench.py --workload $W --streams $S --steps 100 --warmup 20 --no-cpu-baseline --no-extras >> $O/sweep.jsonl 2>> $O/sweep.err
  done
done
python - <<PY
import json
for l in open("$O/sweep.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    c = d["config"]
    print(c["workload"][:10], "streams", c["streams_per_gpu"], "fps", d["value"], "ms", d["ms_per_step"])
PY
tail -3 $O/sweep.err
