"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) as a per-kernel table."""
import csv, glob, os, sqlite3, sys


def rows_from_db(path):
    c = sqlite3.connect(path)
    return list(c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name"))


def rows_from_csv(path, by_grid=None):
    """by_grid: kernels whose name contains this substring are ALSO listed per launch grid (threads in x times y) -- the lock-step tracking
    launches of a run differ in the number of models (grid y), and bench.py quotes the launches of the timed steps only"""
    agg, split = {}, {}
    for r in csv.DictReader(open(path)):
        d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        keys = [(agg, r["Kernel_Name"])]
        if by_grid and by_grid in r["Kernel_Name"]:
            wg = max(1, int(r["Workgroup_Size_X"]))
            keys.append((split, f'{r["Kernel_Name"][:44]} [{int(r["Grid_Size_X"]) // wg} workgroups x {r["Grid_Size_Y"]} models]'))
        for table, k in keys:
            a = table.setdefault(k, [0, 0.0, 1e30, 0.0])
            a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    rows = [(k, v[0], v[1] / v[0], v[2], v[3], v[1]) for k, v in agg.items()]
    return rows, [(k, v[0], v[1] / v[0], v[2], v[3], v[1]) for k, v in split.items()]


def main():
    src = sys.argv[1]
    by_grid = sys.argv[2] if len(sys.argv) > 2 else None   # e.g. icp_reduce
    if os.path.isdir(src):
        cand = glob.glob(os.path.join(src, "**", "*.db"), recursive=True) + glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
        src = cand[0]
    split = []
    if src.endswith(".db"):
        rows = rows_from_db(src)
    else:
        rows, split = rows_from_csv(src, by_grid)
    rows.sort(key=lambda r: -r[5])
    tot = sum(r[5] for r in rows)
    print(f"# source: {os.path.basename(src)}   total kernel time {tot / 1e6:.3f} ms")
    print(f"{'kernel':<72} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'total_ms':>9} {'%':>6}")
    for r in rows:
        print(f"{r[0][:72]:<72} {r[1]:>6} {r[2] / 1e3:>9.2f} {r[3] / 1e3:>9.2f} {r[4] / 1e3:>9.2f} {r[5] / 1e6:>9.3f} {100 * r[5] / tot:>6.2f}")
    if split:
        print(f"# launches of *{by_grid}* by grid")
        for r in sorted(split, key=lambda r: r[0]):
            print(f"{r[0][:100]:<100} {r[1]:>6} {r[2] / 1e3:>9.2f} {r[3] / 1e3:>9.2f} {r[4] / 1e3:>9.2f}")


if __name__ == "__main__":
    main()
