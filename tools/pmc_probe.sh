#!/bin/bash
# PMC probe of the default bench command: FETCH_SIZE pass + an SQ instruction-mix pass; per-kernel averages.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 40 > $O/f.log 2>&1

timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $O/s -o p -- python $R/bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 40 > $O/s.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ("f", "w", "s"):
    files = glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not files:
        print(tag, "no output"); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        k = (r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    with open("$O/%s_summary.txt" % tag, "w") as f:
        for (k, c), (n, s) in sorted(agg.items()):
            f.write("%-50s %-22s n=%5d avg=%14.1f\n" % (k, c, n, s / n))
PY
rm -rf $O/f $O/w $O/s
