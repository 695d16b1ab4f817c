"""Runs the static and the multi-object stream twice each and compares poses / surfel counts bit for bit
(any lost update in the device-side reductions would show up here)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from co_fusion_amd import facade

def run(n_obj, steps):
    cam, frames = bench.make_stream(640, 480, 16, n_obj=n_obj)
    cf = facade.CoFusion(640, 480, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 21, enable_multiple_models=int(n_obj > 0), device_frames_complete=1)
    dev = torch.device("cuda", 0)
    res = [dict(d=torch.from_numpy(f["depth"]).to(dev), c=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
    sig = []
    for i in range(steps):
        k = bench.frame_index(i, 16)
        cf.process_frame_device(res[k]["d"], res[k]["c"], timestamp=i)
        if i % 10 == 9 or i == steps - 1:
            for m in range(cf.num_models):
                info = cf.model_info(m)
                sig.append((i, m, info["count"], info["pose"].tobytes()))
    cf.close()
    return sig

ok = True
for n_obj, steps in ((0, 200), (4, 220)):
    a, b = run(n_obj, steps), run(n_obj, steps)
    same = a == b
    ok &= same
    print(f"n_obj={n_obj}: {len(a)} checkpoints, identical={same}")
sys.exit(0 if ok else 1)
