"""Phases of the fused RGB step + solve launch (diagnostics build: CF_STEP_TRACE, cabi.hip): per tracker, when its workgroups begin, finish
their step, have their atomics acknowledged, hold their ticket, and when the solver ends.  usage: step_trace_summary.py <file>"""
import statistics, sys
from collections import defaultdict
reps = defaultdict(list)
rep = None
for l in open(sys.argv[1]):
    if l.startswith("#"):
        rep = l.strip(); continue
    reps[rep].append([int(x) for x in l.split()])
for rep, rows in reps.items():
    print(rep)
    by = defaultdict(list)
    for r in rows: by[r[1]].append(r)
    for m, v in sorted(by.items()):
        beg = [r[2] for r in v]; step = [r[3] for r in v if r[3] >= 0]; com = [r[4] for r in v if r[4] >= 0]; tic = [r[5] for r in v if r[5] >= 0]
        sol = [r for r in v if r[6] >= 0]
        f = lambda x: f"{min(x) / 1e3:5.2f}/{statistics.median(x) / 1e3:5.2f}/{max(x) / 1e3:5.2f}" if x else "  -  "
        print(f"  tracker {m}: {len(v):4d} wg  begin {f(beg)}  step done {f(step)}  atomics acked {f(com)}  ticket {f(tic)}", end="")
        for r in sol: print(f"  | solver wg {r[0]}: begin {r[2] / 1e3:.2f} ticket {r[5] / 1e3:.2f} solve end {r[6] / 1e3:.2f}", end="")
        print()
