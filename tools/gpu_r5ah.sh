# launch shapes (threads x pixels per lane) and slot orders of the level-0 launch with the LDS accumulators, lines of ONE box
set -u
O=gpurun_out/${1:-r5ah}; mkdir -p $O
timeout 300 python -m pytest tests/test_track_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20"
: > $O/lines.jsonl
for rep in 1 2; do
for W in objects4 static; do
for S in "256 1" "256 2" "512 1" "512 2" "128 1" "128 2" "256 4" "1024 1"; do
  set -- $S
  echo "# $W threads $1 ppt $2" >> $O/lines.jsonl
  timeout 120 $B --workload $W --icp-threads $1 --icp-ppt $2 >> $O/lines.jsonl 2>> $O/err.txt
done; done; done
export CF_LIB_DIR=$PWD/co_fusion_amd/lib_ablate
for rep in 1 2; do
for ORD in 0 1 2 3 4 5 6; do
  echo "# objects4 order $ORD (diagnostics build)" >> $O/lines.jsonl
  CF_ICP_ORDER=$ORD timeout 120 $B >> $O/lines.jsonl 2>> $O/err.txt
done; done
python - <<PY
import json
tag=None
for l in open("$O/lines.jsonl"):
    if l.startswith("#"): tag=l.strip(); continue
    try: d=json.loads(l)
    except Exception: continue
    r=d["roofline"]; print(f"{tag:44s} fps {d['value']:8.2f} icp {r['avg_us']:6.2f} us  digest {d.get('parity_vs_n1',{}).get('sha256','')[:12]}")
PY
