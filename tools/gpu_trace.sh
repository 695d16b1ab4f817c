#!/bin/bash
# kernel timeline of the last frames of a short configs[2] run (for gap analysis): trace csv under gpurun_out/<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python $R/bench.py --workload ${2:-objects4} --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events > $O/trace.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
keep=rows[-6000:]
import gzip
with gzip.open("$O/kernel_trace_tail.csv.gz","wt") as g:
    w=csv.writer(g); w.writerow(["name","start","end","stream","queue"])
    for r in keep: w.writerow([r["Kernel_Name"][:60],r["Start_Timestamp"],r["End_Timestamp"],r.get("Stream_Id",""),r.get("Queue_Id","")])
print(len(rows),"kernels; kept",len(keep))
PY
rm -rf $O/tr
