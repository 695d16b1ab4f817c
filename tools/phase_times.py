"""Host wall-clock per processFrame phase (diagnostics): python tools/phase_times.py [static|objects4|objects4-gt]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import bench
from co_fusion_amd import facade, lib as cflib

wl = sys.argv[1] if len(sys.argv) > 1 else "objects4"
n_obj = 0 if wl == "static" else 4
cam, frames = bench.make_stream(640, 480, 16, n_obj=n_obj)
cf = facade.CoFusion(640, 480, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 21, enable_multiple_models=int(n_obj > 0), device_frames_complete=1)
dev = torch.device("cuda", 0)
res = [dict(d=torch.from_numpy(f["depth"]).to(dev), c=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
host = cflib.load_host()
names = ["prepare", "track", "slic+sums", "unary", "crf", "seg-post", "model-logic", "fuse+clean", "predict"]
out = (C.c_double * 9)(); fr = C.c_long()
def run(lo, hi):
    for i in range(lo, hi):
        k = bench.frame_index(i, 16)
        if wl == "objects4-gt":
            f = frames[k]; cf.process_frame(f["depth"], f["rgb"], mask=(f["label"] * 40).astype(np.uint8), timestamp=i)
        else:
            cf.process_frame_device(res[k]["d"], res[k]["c"], timestamp=i)
run(0, 150)
host.cofusion_debug_phase_ms(out, 9, C.byref(fr), 1)
import time
torch.cuda.synchronize(); t0 = time.perf_counter(); run(150, 250); torch.cuda.synchronize(); dt = time.perf_counter() - t0
host.cofusion_debug_phase_ms(out, 9, C.byref(fr), 0)
print(f"{wl}: {1e3 * dt / 100:.3f} ms/frame wall, models {cf.num_models}")
for n, v in zip(names, out):
    print(f"  {n:<12} {v / max(1, fr.value):8.3f} ms/frame")
