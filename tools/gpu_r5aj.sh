set -u
O=gpurun_out/${1:-r5aj}; mkdir -p $O
timeout 300 python -m pytest tests/test_track_gpu.py tests/test_refpin_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -1 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20"
: > $O/lines.jsonl
for rep in 1 2 3; do
for W in objects4 static; do
for S in "lib 1" "lib 2" "lib_old 1"; do
  set -- $S
  echo "# $W $1 ppt $2" >> $O/lines.jsonl
  CF_LIB_DIR=$PWD/co_fusion_amd/$1 timeout 120 $B --workload $W --icp-ppt $2 >> $O/lines.jsonl 2>> $O/err.txt
done; done; done
export CF_LIB_DIR=$PWD/co_fusion_amd/lib_ablate
CF_ICP_TRACE=185 CF_ICP_TRACE_OUT=$O/trace.txt timeout 150 $B --icp-ppt 1 > /dev/null 2>> $O/err.txt
python tools/icp_trace_summary.py $O/trace.txt > $O/trace_summary.txt 2>&1
python - <<PY
import json
tag=None
for l in open("$O/lines.jsonl"):
    if l.startswith("#"): tag=l.strip(); continue
    try: d=json.loads(l)
    except Exception: continue
    r=d["roofline"]; print(f"{tag:34s} fps {d['value']:8.2f} icp {r['avg_us']:6.2f} us  frac {r['frac']:.4f} digest {d.get('parity_vs_n1',{}).get('sha256','')[:12]}")
PY
head -14 $O/trace_summary.txt
