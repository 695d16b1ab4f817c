"""Generates tests/golden/ref_full_v1.npz: the two reference pins (tracking kernels: tests/refpin.py run(); surfel shaders:
surfel_run()) at BASELINE.json's frame size, 640x480.  Large arrays are stored as sha256 digests (planar maps after blanking the
y/z planes the reference leaves undefined where x is NaN), the normal-equation results verbatim, and the inputs as digests only:
they are regenerated from the seeded generator at test time and checked against the digests.
Run from the repo root (needs /root/reference, about two minutes):  python tests/golden/make_ref_full_golden.py"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore", category=RuntimeWarning)
W, H = 640, 480
PLANAR = ("vmap", "nmap", "copy_", "resize_", "model_v", "model_n")


def planar_valid_only(m):
    m = m.copy()
    h = m.shape[0] // 3
    bad = np.isnan(m[:h])
    m[h:2 * h][bad] = 0; m[2 * h:][bad] = 0
    return m


if __name__ == "__main__":
    import orc
    import ref
    import refpin
    assert ref.available(), "needs /root/reference (oracle/ref_shim/build_ref.py)"
    blob = {"size": np.array([W, H], np.int64)}
    inp = refpin.inputs(W, H)
    for k, v in inp.items():
        blob["insha_" + k] = refpin.digest(v)
    out = refpin.run(ref, inp, orc.Cam)
    for k, v in out.items():
        v = np.asarray(v)
        if v.size <= 64:
            blob["val_" + k] = v
        else:
            blob["sha_" + k] = refpin.digest(planar_valid_only(v) if k.startswith(PLANAR) else v)
            blob["dtype_" + k] = np.array(str(v.dtype))
    sinp = refpin.surfel_inputs(W, H)
    for k, v in sinp.items():
        blob["sinsha_" + k] = refpin.digest(v)
    with ref.surfel_passes() as op:
        sout = refpin.surfel_run(refpin.CpuSurfelBackend(op, sinp["cam"]), sinp)
    for k, v in sout.items():
        blob["ssha_" + k] = refpin.digest(v)
    blob["ssummary"] = refpin.surfel_summary(sout)
    path = os.path.join(HERE, "ref_full_v1.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes;", len(out), "+", len(sout), "pinned arrays; surfel summary", blob["ssummary"])
    print({k: out[k] for k in ("icp_res0", "residual_sigma_count0", "so3_res")})
