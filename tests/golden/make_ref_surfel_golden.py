"""Generates tests/golden/ref_surfel_v1.npz: sha256 digests (+ sizes) of what the REFERENCE'S OWN surfel shaders
(Core/Shaders/*.vert / *.frag, compiled to C++ by oracle/ref_shim/build_ref.py and driven by oracle/ref_shim/ref_gl.cpp) produce
on the scripted scenario of tests/refpin.py (surfel_run).  Inputs are stored verbatim next to the digests.
Run from the repo root (needs /root/reference):  python tests/golden/make_ref_surfel_golden.py"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore", category=RuntimeWarning)

if __name__ == "__main__":
    import ref
    import refpin
    assert ref.available(), "needs /root/reference (oracle/ref_shim/build_ref.py)"
    inp = refpin.surfel_inputs()
    with ref.surfel_passes() as op:
        out = refpin.surfel_run(refpin.CpuSurfelBackend(op, inp["cam"]), inp)
    blob = {"in_" + k: v for k, v in inp.items()}
    for k, v in out.items():
        blob["sha_" + k] = refpin.digest(v)
        blob["shape_" + k] = np.array(np.asarray(v).shape, np.int64)
    blob["summary"] = refpin.surfel_summary(out)
    path = os.path.join(HERE, "ref_surfel_v1.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes,", len(out), "pinned arrays; summary", blob["summary"])
