#!/usr/bin/env python
"""Run a .klg RGB-D log through the hot path and export the reference's outputs (what `CoFusion -l <log> -exportdir <dir>`
does head-less, GUI/MainController.cpp):

    python tools/run_klg.py seq.klg out/ [--static] [--width 640 --height 480 --fx 528 --fy 528 --cx 320 --cy 240]
                            [--frames N] [--flip-colors] [--export-segmentation]

Writes out/poses-<id>.txt, out/cloud-<id>.ply (and out/Segmentation<tick>.png) and prints frames/s."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("log"); ap.add_argument("outdir")
    ap.add_argument("--static", action="store_true", help="single static model (-static)")
    ap.add_argument("--width", type=int, default=640); ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--fx", type=float, default=528.0); ap.add_argument("--fy", type=float, default=528.0)
    ap.add_argument("--cx", type=float, default=320.0); ap.add_argument("--cy", type=float, default=240.0)
    ap.add_argument("--frames", type=int, default=-1)
    ap.add_argument("--flip-colors", action="store_true")
    ap.add_argument("--export-segmentation", action="store_true")
    ap.add_argument("--max-surfels", type=int, default=3072 * 3072)
    a = ap.parse_args()
    from co_fusion_amd import facade, klg
    os.makedirs(a.outdir, exist_ok=True)
    prefix = a.outdir.rstrip("/") + "/"
    log = klg.KlgReader(a.log, a.width, a.height, flip_colors=a.flip_colors)
    cf = facade.CoFusion(a.width, a.height, a.fx, a.fy, a.cx, a.cy, max_surfels=a.max_surfels, enable_multiple_models=int(not a.static),
                         enable_pose_logging=1)
    if a.export_segmentation and not a.static:
        cf.set_export_segmentation(prefix)
    n, t0 = 0, time.perf_counter()
    for ts, depth, rgb in log:
        cf.process_frame(depth, rgb, timestamp=ts)
        n += 1
        if 0 < a.frames <= n:
            break
    dt = time.perf_counter() - t0
    print(f"{n} frames of {log.num_frames} in {dt:.2f} s ({n / dt:.1f} frames/s incl. log decoding and upload), {cf.num_models} active models")
    print(f"exported {cf.export_poses(prefix)} pose file(s), {cf.save_ply(prefix)} PLY cloud(s) to {prefix}")
    cf.close()


if __name__ == "__main__":
    main()
