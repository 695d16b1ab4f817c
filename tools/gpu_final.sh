#!/bin/bash
# final numbers of a round: rocprofv3 kernel stats of the default bench command, then the bench lines of the other configurations
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/${1:-r05d}; mkdir -p $out; : > $out/lines.jsonl
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python $R/bench.py > $out/bench_default_line.json 2> $out/prof.log
python $R/tools/prof_summary.py $out/prof > $out/bench_objects4_kernel_stats.txt 2>&1
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/bench_objects4_rocprofv3_kernel_stats.csv \;
rm -rf $out/prof
cd $R
run() { echo "== $*" >> $out/log.txt; timeout 300 python bench.py "$@" 2>> $out/log.txt | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); d['args'] = '$*'; print(json.dumps(d))
except Exception as e:
    print(json.dumps({'args': '$*', 'error': str(e), 'raw': l[:200]}))" >> $out/lines.jsonl; }
run --no-cpu-baseline --no-extras --workload static
run --no-cpu-baseline --no-extras --workload objects8
run --no-cpu-baseline --no-extras --workload big --steps 40 --warmup 10
run --no-cpu-baseline --no-extras --workload big-static --steps 60 --warmup 10
run --no-cpu-baseline --no-extras --workload objects4 --streams 3 --lockstep
run --no-cpu-baseline --no-extras --workload static --streams 4 --lockstep
tail -1 $out/bench_default_line.json | cut -c1-600
grep -E "icp_reduce|gn_solve|rgb_slot|so3_prealign|model_maps|seg_unary" $out/bench_objects4_kernel_stats.txt | cut -c1-150 | head -12
python - <<PY
import json
for l in open('$out/lines.jsonl'):
    d = json.loads(l); r = d.get('roofline', {})
    print(d.get('args'), '| fps', d.get('value'), '| ms', d.get('ms_per_step'), '| icp us', r.get('avg_us'), 'frac', r.get('frac'), d.get('error'))
PY
