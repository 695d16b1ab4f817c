#!/usr/bin/env python3
"""tests/golden/ref_odo_v1.npz: what the REFERENCE'S OWN RGBDOdometry class (Core/Utils/RGBDOdometry.{h,cpp} + OdometryProvider.h,
compiled from /root/reference by oracle/ref_shim/build_ref.py: CUDA kernels under the CPU SIMT emulator, Eigen / GPUTexture /
Stopwatch stand-ins) returns for recorded tracking inputs of the oracle's -static frame loop, for six option sets of
getIncrementalTransformation.  Needs /root/reference; about 15 minutes (the emulator runs every CUDA thread as a fiber).
The inputs are not stored: tests re-record them (0.8 s) and check their digest."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refodo  # noqa: E402

W, H, N_FRAMES, FRAMES = 160, 120, 5, (1, 3)   # tracked frames 1 and 3 of a 5-frame run (0-based among the tracked ones)
# `make_ref_odo_golden.py full`: the same pin at BASELINE.json's own frame size, 640x480 (two frames x {default, fast_odom}; the emulator
# needs ~20 minutes per call there) -> ref_odo_full_v1.npz
FULL = len(sys.argv) > 1 and sys.argv[1] == "full"
if FULL:
    W, H = 640, 480
OUT_NAME = "ref_odo_full_v1.npz" if FULL else "ref_odo_v1.npz"


def digest(fr):
    h = hashlib.sha256()
    for k in ("prev_rgba", "v4", "n4", "pose", "img", "rgba"):
        h.update(np.ascontiguousarray(fr[k]).tobytes())
    for d in fr["depth_pyr"]:
        h.update(np.ascontiguousarray(d).tobytes())
    return h.hexdigest()


def main():
    cam, frames = refodo.record_tracking_inputs(W, H, N_FRAMES)
    out = dict(meta=np.array([W, H, N_FRAMES], np.int32), frames=np.array(FRAMES, np.int32),
               options=np.array([o[0] for o in refodo.OPTION_SETS if not FULL or o[0] in ("default", "fast_odom")]))
    for fi in FRAMES:
        fr = frames[fi]
        out[f"f{fi}/digest"] = np.array(digest(fr))
        for opts in ([o for o in refodo.OPTION_SETS if o[0] in ("default", "fast_odom")] if FULL else refodo.OPTION_SETS):
            tr, rot, st, err = refodo.track_once(refodo.RefOdometry, cam, W, H, fr, opts)
            key = f"f{fi}/{opts[0]}"
            out[key + "/trans"] = tr; out[key + "/rot"] = rot
            out[key + "/stats"] = np.array([st["last_icp_error"], st["last_icp_count"], st["last_rgb_error"], st["last_rgb_count"],
                                            st["last_so3_error"], st["last_so3_count"]], np.float32)
            out[key + "/lastA"] = st["lastA"]; out[key + "/lastb"] = st["lastb"]
            out[key + "/err_sum_max"] = np.array([err.astype(np.float64).sum(), err.max()], np.float64)
            print(key, tr, st["last_icp_count"], st["last_rgb_count"], flush=True)
    np.savez_compressed(os.path.join(HERE, OUT_NAME), **out)


if __name__ == "__main__":
    main()
