#!/bin/bash
# LDLT phase of the solve kernel for diagnostics builds that differ in CF_LDLT_SHUFFLE_STEPS (cf_device.h): co_fusion_amd/lib_ls<N>, lib_ablate
#   usage: gpu_ldlt_variants.sh <outdir> <libdir names...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-ldlt}; mkdir -p $O; shift
cd $R
for rep in 1 2; do
for L in "$@"; do
  CF_LIB_DIR=$R/co_fusion_amd/$L CF_SOLVE_TRACE=150 CF_ICP_TRACE_OUT=$O/trace_$L.txt timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 10 > /dev/null 2> $O/err_$L.txt
  python - <<PY
import statistics
rows = [[int(x) for x in l.split()] for l in open("$O/trace_$L.txt") if not l.startswith("#")]
ph = lambda a, b: statistics.median([r[1 + b] - r[1 + a] for r in rows])
print("%-12s solves %2d  LDLT phase %5.0f ns   end of write-back %5.0f ns" % ("$L", len(rows), ph(2, 3), statistics.median([r[8] for r in rows])))
PY
done; done | tee -a $O/ldlt_variants.txt
