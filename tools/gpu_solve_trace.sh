#!/bin/bash
# phases of the Gauss-Newton solve kernel (diagnostics build, CF_SOLVE_TRACE: tracker 0's solves of one tracking call) + the kernel's
# rocprofv3 duration in the default bench command.   usage: gpu_solve_trace.sh <outdir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-solve}
mkdir -p $O
export TMPDIR=/tmp
cd $R
CF_LIB_DIR=$R/co_fusion_amd/lib_ablate CF_SOLVE_TRACE=${CALL:-150} CF_ICP_TRACE_OUT=$O/solve_trace.txt timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $O/bench_ablate.json 2> $O/bench_ablate.err
python - <<PY
import statistics
rows = [[int(x) for x in l.split()] for l in open("$O/solve_trace.txt") if not l.startswith("#")]
names = ["loaded", "totals", "unpacked", "LDLT", "rodrigues", "pose", "next-it", "write-back"]
print("solves traced:", len(rows))
prev = [0] * len(rows)
for k, n in enumerate(names):
    col = [r[1 + k] for r in rows]
    d = [c - p for c, p in zip(col, prev)]
    print(f"  {n:10s} at {statistics.median(col):7.0f} ns (median), phase {statistics.median(d):6.0f} ns")
    prev = col
PY
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python $R/bench.py --no-cpu-baseline --no-extras > $O/prof.log 2>&1
python $R/tools/prof_summary.py $O/prof > $O/kernel_stats_objects4.txt 2>&1; head -${PROF_LINES:-14} $O/kernel_stats_objects4.txt
rm -rf $O/prof
tail -c 400 $O/bench_ablate.json
