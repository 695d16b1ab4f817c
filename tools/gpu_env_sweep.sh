#!/bin/bash
# bench lines of one workload under a list of environment settings, all on ONE box (boxes differ by +-10 %)
#   usage: gpu_env_sweep.sh <outdir> <workload> "<ENV=V ENV2=V>" "<...>" ...     ("-" = no extra environment)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-sweep}
W=${2:-objects4}
shift 2
mkdir -p $O
cd $R
export TMPDIR=/tmp
: > $O/sweep.jsonl
for rep in $(seq 1 ${REPS:-1}); do
  for E in "$@"; do
    [ "$E" = "-" ] && E=""
    env $E timeout 150 python bench.py --workload $W --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $O/line.json 2>> $O/sweep.err
    python - "$E" $O/line.json <<'PY' | tee -a $O/sweep.txt
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{sys.argv[1] or '-':40s} fps {d['value']:8.1f}  ms {d['ms_per_step']:.4f}  icp L0 us {r['avg_us']:.2f}  models {d['config']['active_models']}")
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
    cat $O/line.json >> $O/sweep.jsonl
  done
done
