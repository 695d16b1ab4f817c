set -u
O=gpurun_out/r5af; mkdir -p $O
export CF_LIB_DIR=$PWD/co_fusion_amd/lib_ablate
B="python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20"
: > $O/lines.jsonl
for rep in 1 2; do
for v in "CF_DUMMY=1" "CF_NO_OCC=1" "CF_NO_ZCULL=1" "CF_NO_OCC=1 CF_NO_ZCULL=1"; do
  echo "# $v" >> $O/lines.jsonl
  env $v timeout 120 $B >> $O/lines.jsonl 2>> $O/err.txt
done; done
CF_ICP_REPLAY=180 timeout 150 $B > /dev/null 2> $O/replay.txt
CF_ICP_TRACE=185 CF_ICP_TRACE_OUT=$O/trace.txt timeout 150 $B > /dev/null 2>> $O/err.txt
python tools/icp_trace_summary.py $O/trace.txt > $O/trace_summary.txt 2>&1
python - <<PY
import json
tag=None
for l in open("$O/lines.jsonl"):
    if l.startswith("#"): tag=l.strip(); continue
    try: d=json.loads(l)
    except Exception: continue
    r=d["roofline"]; print(f"{tag:32s} fps {d['value']:8.2f} icp {r['avg_us']:6.2f} us")
PY
grep "icp replay" $O/replay.txt
head -16 $O/trace_summary.txt
