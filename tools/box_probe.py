"""Diagnostics: the screen boxes (cf_track_stats::cull_box) of the object models on the headline workload."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from co_fusion_amd import facade

W, H = 640, 480
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cam, frames = bench.make_stream(W, H, 16, n_obj=n_obj)
cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 21, enable_multiple_models=1)
P = 24 * n_obj + 30
for i in range(P + 20):
    f = frames[bench.frame_index(i, 16)]
    if i < P:
        cf.process_frame(f["depth"], f["rgb"], mask=(f["label"] * 40).astype(np.uint8), timestamp=i)
    else:
        cf.process_frame(f["depth"], f["rgb"], timestamp=i)
    if i >= P + 16:
        out = []
        for m in range(cf.num_models):
            b = cf.model_cull_box(m)
            x0, y0, x1, y1 = max(b[0], 0), max(b[1], 0), min(b[2], W - 1), min(b[3], H - 1)
            area = max(0, x1 - x0 + 1) * max(0, y1 - y0 + 1)
            lab = frames[bench.frame_index(i, 16)]["label"]
            out.append((b, round(area / (W * H), 3)))
        print(i, out, "gt px per label:", [int((lab == k).sum()) for k in range(1, n_obj + 1)], flush=True)
cf.close()
