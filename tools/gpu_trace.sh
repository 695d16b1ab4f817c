#!/bin/bash
# kernel trace (timestamps) of a short default bench run, kept as CSV for timeline analysis (tools/frame_timeline.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-30} --warmup 10 ${BENCH_ARGS:-} > $O/trace.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keep = rows[-${KEEP:-6000}:]
w = csv.writer(open("$O/kernel_trace_tail.csv", "w"))
w.writerow(["name", "start_ns", "end_ns", "queue", "grid_x", "wg_x"])
t0 = int(keep[0]["Start_Timestamp"])
for r in keep:
    w.writerow([r["Kernel_Name"][:60], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r.get("Queue_Id", ""), r["Grid_Size_X"], r["Workgroup_Size_X"]])
print("kept", len(keep), "of", len(rows))
PY
rm -rf $O/tr
tail -2 $O/trace.log
