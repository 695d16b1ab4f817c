"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv), optionally filtered by grid size.

usage: pmc_summary.py <dir-or-csv> [kernel-substring] [grid-size]
"""
import csv, glob, os, sys


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    grid = int(sys.argv[3]) if len(sys.argv) > 3 else None
    if os.path.isdir(src):
        src = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(src)):
        if pat not in r["Kernel_Name"]:
            continue
        if grid is not None and int(r["Grid_Size"]) != grid:
            continue
        key = (r["Kernel_Name"][:64], int(r["Grid_Size"]), r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    print(f"# source: {os.path.basename(src)}")
    print(f"{'kernel':<64} {'grid':>9} {'counter':<24} {'dispatches':>10} {'avg':>16}")
    for (k, g, c), (n, s) in sorted(agg.items()):
        print(f"{k:<64} {g:>9} {c:<24} {n:>10} {s / n:>16.2f}")


if __name__ == "__main__":
    main()
