"""Summary of a per-workgroup trace of the level-0 {ICP || residual} launch (diagnostics build: CF_ICP_TRACE / CF_ICP_TRACE_OUT, cabi.hip).
usage: icp_trace_summary.py <trace.txt>"""
import statistics, sys
from collections import defaultdict

rows = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1]) if not l.startswith("#")]
print(open(sys.argv[1]).readline().strip())
kinds = {0: "culled ICP", 1: "unculled ICP", 2: "residual"}
by = defaultdict(list)
for b, kind, model, t0, t1, xcc, hwid in rows:
    by[(kind, model)].append((t0, t1, b))
end = max(r[4] for r in rows)
print(f"launch: first begin 0, last end {end / 1e3:.2f} us, {len(rows)} workgroups")
for (kind, model), v in sorted(by.items()):
    s = [a for a, _, _ in v]; e = [b for _, b, _ in v]; d = [b - a for a, b, _ in v]
    print(f"  {kinds[kind]:13s} model {model}: {len(v):5d} wg  begin {min(s) / 1e3:5.2f}/{statistics.median(s) / 1e3:5.2f}/{max(s) / 1e3:5.2f}  "
          f"duration {min(d) / 1e3:5.2f}/{statistics.median(d) / 1e3:5.2f}/{max(d) / 1e3:5.2f}  end max {max(e) / 1e3:5.2f} us")
step = 500
print("active workgroups every 0.5 us (begun / running / by kind running):")
for t in range(0, end + step, step):
    run = [r for r in rows if r[3] <= t < r[4]]
    k = [sum(1 for r in run if r[1] == q) for q in (0, 1, 2)]
    print(f"  {t / 1e3:5.1f} us  begun {sum(1 for r in rows if r[3] <= t):5d}  running {len(run):5d}  culled/unculled/residual {k[0]:4d}/{k[1]:4d}/{k[2]:4d}")
late = sorted(rows, key=lambda r: -r[4])[:12]
print("last to end:")
for b, kind, model, t0, t1, xcc, hwid in late:
    print(f"  wg {b:5d} {kinds[kind]:13s} model {model} begin {t0 / 1e3:5.2f} end {t1 / 1e3:5.2f} xcc {xcc}")
