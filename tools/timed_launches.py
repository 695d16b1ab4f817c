"""rocprofv3's own durations of the LAST n dispatches of a kernel in a --kernel-trace run: the level-0 {ICP || residual} launches of the
timed steps of `bench.py --steps K` are the last 10 K dispatches of the multi-tracker level-0 kernel (tag 4), whatever ran in the pre-roll.
usage: timed_launches.py <dir or *_kernel_trace.csv> <kernel name substring> <n> [algorithmic bytes per launch]"""
import csv, glob, os, statistics, sys

src, sub, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
nbytes = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [(float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(src))
        if sub in r["Kernel_Name"]]
rows.sort()
names = sorted({r[2] for r in rows})
d = [r[1] / 1e3 for r in rows[-n:]]
print(f"{len(rows)} dispatches of {names} in the run; the last {len(d)}:")
print(f"  mean {statistics.mean(d):.2f}  median {statistics.median(d):.2f}  min {min(d):.2f}  max {max(d):.2f} us")
if nbytes:
    print(f"  -> {nbytes:.0f} algorithmic bytes / {statistics.mean(d):.2f} us = {nbytes / statistics.mean(d) / 1e6:.3f} TB/s = {nbytes / statistics.mean(d) / 1e6 / 8:.3f} of 8 TB/s")
