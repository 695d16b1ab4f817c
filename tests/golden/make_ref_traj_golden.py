"""Trajectory-level pin against the reference's OWN arithmetic: the pinned frame loop (the text of
CoFusion::processFrame, oracle/ref_shim) is played with the reference's own RGBDOdometry class as the tracker of every model -- its CUDA
kernels under the CPU emulator, f32 tree reductions (reduce.cu:90-185), Eigen-style host solve (RGBDOdometry.cpp:217-477) -- and the
poses of every frame are stored.  tests/test_cpu_refpin.py plays the same streams with the oracle's exact-integer tracker (the bits the
HIP path reproduces, tests/test_configs_gpu.py) and bounds the trajectory difference (ATE).

The emulator needs about a minute per tracked model and frame, so this runs once here (CPU container) and the result is committed:

    python tests/golden/make_ref_traj_golden.py [scenario ...]      ->  tests/golden/ref_traj_v1.npz
"""
from __future__ import annotations

import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore", category=RuntimeWarning)

OUT = os.path.join(HERE, "ref_traj_v1.npz")
W, H = 160, 128
MAXM = 6
# name -> (objects in the scene, frames, conf_global, spawn offset, multiple models)
SCENARIOS = {
    "static_camera": (0, 40, 10.0, 20, False),          # `-static`: one model, the camera trajectory
    "crf_two_objects": (2, 24, 0.5, 3, True),           # motion CRF, models spawn on the way: camera and object poses
}


def play(name, reference_tracker, n_frames=None, log=None):
    """poses [F, MAXM, 4, 4], ids [F, MAXM] (-1: no model), counts [F, MAXM] of the pinned frame loop"""
    import refcofusion
    from co_fusion_amd import synth
    n_obj, frames, conf_global, spawn, multi = SCENARIOS[name]
    F = n_frames or frames
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=n_obj)
    cf = refcofusion.RefCoFusion(cam, conf_global=conf_global, spawn_offset=spawn, multi=multi, reference_tracker=reference_tracker)
    poses = np.zeros((F, MAXM, 4, 4), np.float32); ids = np.full((F, MAXM), -1, np.int32); counts = np.zeros((F, MAXM), np.int64)
    t0 = time.time()
    for t in range(F):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        cf.process_frame(d, rgb, timestamp=t)
        for i in range(min(cf.num_models, MAXM)):
            m = cf.model(i)
            poses[t, i] = m["pose"]; ids[t, i] = m["id"]; counts[t, i] = m["count"]
        if log:
            log(f"{name} frame {t}: {time.time() - t0:.0f} s, ids {ids[t][ids[t] >= 0].tolist()}, camera {poses[t, 0, :3, 3]}")
    return poses, ids, counts


def main():
    names = sys.argv[1:] or list(SCENARIOS)
    data = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        poses, ids, counts = play(name, True, log=lambda s: print(s, flush=True))
        data[f"{name}/poses"] = poses; data[f"{name}/ids"] = ids; data[f"{name}/counts"] = counts
        np.savez_compressed(OUT, **data)
    print("wrote", OUT)


if __name__ == "__main__":
    # one scenario per process: Core/Segmentation keeps function-static state (tests/cfpin.py: run_reference_isolated)
    if len(sys.argv) > 2:
        import subprocess
        for n in sys.argv[1:]:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), n])
    else:
        main()
