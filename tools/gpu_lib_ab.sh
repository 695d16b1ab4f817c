#!/bin/bash
# A/B of two builds on ONE box: co_fusion_amd/lib against co_fusion_amd/lib_old (another source state built aside with
# `make LIBDIR=.../lib_old` in csrc/ and host/, loaded through CF_LIB_DIR); chosen tests first, then alternating bench lines and the
# rocprofv3 kernel statistics of both
#   usage: gpu_lib_ab.sh <outdir> "<tests>" "<kernel name pattern for the statistics>" [workloads...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ab}; mkdir -p $O
T=${2:-tests/test_track_gpu.py}
PAT=${3:-icp_reduce}
shift 3
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest $T -m gpu -x -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -2 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-extras --steps 150 --warmup 20"
: > $O/lines.jsonl
for rep in 1 2 3; do
for W in ${@:-objects4 static}; do
for L in lib lib_old; do
  echo "# $W $L" >> $O/lines.jsonl
  CF_LIB_DIR=$R/co_fusion_amd/$L timeout 200 $B --workload $W >> $O/lines.jsonl 2>> $O/err.txt
done; done; done
python - <<PY
import json
tag=None
for l in open("$O/lines.jsonl"):
    if l.startswith("#"): tag=l.strip(); continue
    try: d=json.loads(l)
    except Exception: continue
    r=d["roofline"]; print(f"{tag:34s} fps {d['value']:8.2f}  ms {d['ms_per_step']:.4f}  icp {r['avg_us']:6.2f} us  digest {d.get('parity_vs_n1',{}).get('sha256','')[:12]}")
PY
cd /tmp
for L in lib lib_old; do
CF_LIB_DIR=$R/co_fusion_amd/$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$L -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 20 > /dev/null 2> $O/prof_$L.log
echo "--- $L"; python $R/tools/prof_summary.py $O/prof_$L 2>&1 | grep -E "$PAT|kernel  " | tee -a $O/kernel_stats_$L.txt
rm -rf $O/prof_$L
done
