"""Launch-shape sweep of the ICP reduction (micro-benchmark; run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
from co_fusion_amd import api

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
fp = common.frame_pair(W, H)
cam = fp["cam"]
ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
g = api.Odometry(ctx)
d = ctx.to_device
pose = common.perturbed_pose(2)
g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(fp["v4"]), d(fp["n4"]), pose); g.init_rgb_model(d(fp["img"]))
g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
g.track(pose[:3, 3], pose[:3, :3])
for threads, ppt in [(256, 1), (256, 2), (256, 4), (128, 2), (128, 4), (512, 1), (512, 2), (1024, 1), (64, 4)]:
    ctx.set_icp_launch(threads, ppt)
    res = []
    for lvl in range(3):
        us = g.bench_icp(lvl, 300)
        n = (W >> lvl) * (H >> lvl)
        res.append(f"L{lvl}: {us:7.2f} us {48 * n / us / 1e3:8.1f} GB/s")
    print(f"threads={threads:4d} ppt={ppt}  " + "  ".join(res))
