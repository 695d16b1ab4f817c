"""profiles/rNN_icp_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/gpu_pmc.sh, stamped with the sha256 of the kernel's
source file so that bench.py only quotes it for the build it was taken on.

usage: make_traffic_json.py <gpurun_out/dir with pmc_icp_FETCH_SIZE.txt + pmc_icp_WRITE_SIZE.txt> <profiles/rNN_icp_traffic.json> [models]
The level-0 multi-model kernel is icp_reduce_kernel<P, 4, ...> (P pixels per lane: 2 since round 5), the one-model one <P, 0, ...>; with the box-indexed grids of round 4 the
grid size of the multi-model launch varies from frame to frame, so all its dispatches with `models` trackers' worth of residual workgroups
are averaged (weighted by dispatches)."""
import hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def averages(path, tag):
    n, s = 0, 0.0
    for l in open(path):
        if not re.search(tag, l):
            continue
        parts = l.split()
        d, avg = int(parts[-2]), float(parts[-1])
        n += d; s += d * avg
    return (s / n if n else None), n


def main():
    src, out = sys.argv[1], sys.argv[2]
    # optional: rocprofv3 --kernel-trace durations of the same launches on the same build ("objects4=12.48,static=9.64": average us of the
    # level-0 launches of the timed steps, profiles/rNN_icp_level0_timed_launches.txt) -- bench.py quotes them beside its own events
    rocprof = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 else {}
    sha = hashlib.sha256(open(os.path.join(ROOT, "co_fusion_amd", "csrc", "track_reduce.hip"), "rb").read()).hexdigest()
    entries = []
    for tag, workload, what, stem, pixels in (
            (r"icp_reduce_kernel<\d, 4, false>", "objects4", "ICP reduction of the lock-step models || their RGB residual passes, level 0, box-indexed grids", "pmc_icp", 307200),
            (r"icp_reduce_kernel<\d, 0, false>", "static", "ICP reduction || RGB residual of the background model, level 0 (the pre-roll frames of the same run)", "pmc_icp", 307200),
            (r"icp_reduce_kernel<\d, 0, false>", "big-static", "ICP reduction || RGB residual of one static model at 1280x960, level 0 (round 6: a counter pass of its own)", "pmc_icp_big_static", 1228800)):
        if not os.path.exists(os.path.join(src, stem + "_FETCH_SIZE.txt")):
            continue
        f, nf = averages(os.path.join(src, stem + "_FETCH_SIZE.txt"), tag)
        w, nw = averages(os.path.join(src, stem + "_WRITE_SIZE.txt"), tag)
        if f is None or w is None:
            continue
        entries.append(dict(kernel=f"cf::{tag.replace(chr(92) + 'd', 'P')}: {what}", workload=workload, pixels=pixels, fetch_size_kb_avg=round(f, 2), write_size_kb_avg=round(w, 2),
                            dispatches=min(nf, nw), kernel_source_sha256=sha,
                            correction="traffic = 2*FETCH_SIZE*1024 + 1*WRITE_SIZE*1024 (factors measured by tools/microbench/fetch_calib.hip in the same call: pmc_calibration_*.txt)",
                            traffic_bytes_per_launch=int(2 * f * 1024 + w * 1024),
                            **({"rocprofv3_avg_us": rocprof[workload]} if workload in rocprof else {}),
                            source=f"{os.path.basename(src.rstrip('/'))}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py [--workload big-static] --no-cpu-baseline --no-extras --steps 20 --warmup 5` (tools/gpu_r6_measure.sh)"))
    json.dump(entries, open(out, "w"), indent=1)
    print(json.dumps(entries, indent=1))


if __name__ == "__main__":
    main()
