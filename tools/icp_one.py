"""Runs the ICP reduction micro-benchmark for one launch shape (for rocprofv3 PMC passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from co_fusion_amd import api

threads, ppt, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
W, H = 640, 480
fp = common.frame_pair(W, H)
cam = fp["cam"]
ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
g = api.Odometry(ctx)
d = ctx.to_device
pose = common.perturbed_pose(2)
g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(fp["v4"]), d(fp["n4"]), pose); g.init_rgb_model(d(fp["img"]))
g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
g.track(pose[:3, 3], pose[:3, :3])
ctx.set_icp_launch(threads, ppt)
print("L0 us", g.bench_icp(0, iters))
