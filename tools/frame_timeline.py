"""Timeline of one steady-state frame from gpurun_out/<tag>/kernel_trace_tail.csv (tools/gpu_trace.sh): phases, gaps, the kernels in order.
usage: frame_timeline.py <csv> [frame-index] [--list]"""
import csv, statistics, sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    for r in rows:
        r["s"] = int(r["start_ns"]); r["e"] = int(r["end_ns"])
    so3 = [i for i, r in enumerate(rows) if "so3_prealign" in r["name"]]
    per = []
    for a, b in zip(so3[3:-1], so3[4:]):
        fr = rows[a:b]
        iv = sorted((r["s"], r["e"]) for r in fr)
        busy, (cs, ce) = 0, iv[0]
        for s, e in iv[1:]:
            if s > ce:
                busy += ce - cs; cs, ce = s, e
            else:
                ce = max(ce, e)
        per.append((rows[b]["s"] - fr[0]["s"], busy + ce - cs, len(fr)))
    print(f"frames {len(per)}: period median {statistics.median(p[0] for p in per) / 1e3:.1f} us, GPU busy (union) {statistics.median(p[1] for p in per) / 1e3:.1f} us, "
          f"kernels per frame {statistics.median(p[2] for p in per)}")
    k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(so3) // 2
    a, b = so3[k], so3[k + 1]
    fr = sorted(rows[a:b], key=lambda r: r["s"]); t0 = fr[0]["s"]

    def span(name):
        xs = [r for r in fr if name in r["name"]]
        return ((xs[0]["s"] - t0) / 1e3, (max(r["e"] for r in xs) - t0) / 1e3, len(xs)) if xs else None
    for nm in ("so3_prealign", "icp_reduce", "gn_solve", "seg_accumulate", "seg_unary", "crf_", "seg_post", "seg_upsample", "index_splat", "associate", "update_kernel",
               "clean_kernel", "splat_raster", "splat_resolve", "fill_in", "bilateral", "model_maps", "rgbd_", "frame_maps", "rgb_prep", "slic_"):
        sp = span(nm)
        if sp:
            print(f"  {nm:16s} {sp[0]:8.1f} .. {sp[1]:8.1f} us  ({sp[2]} launches)")
    ce, last = fr[0]["e"], fr[0]
    for r in fr[1:]:
        if r["s"] - ce > 8000:
            print(f"  gap {(r['s'] - ce) / 1e3:6.1f} us at {(ce - t0) / 1e3:8.1f}: after {last['name'][4:44]} -> {r['name'][4:44]}")
        if r["e"] > ce:
            ce, last = r["e"], r
    if "--list" in sys.argv:
        for r in fr:
            print(f"{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:6.1f}  q{r['queue']:>3} wg {int(r['grid_x']) // max(1, int(r['wg_x'])):>6}  {r['name'][:60]}")


if __name__ == "__main__":
    main()
