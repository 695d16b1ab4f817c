"""Generates tests/golden/ref_v1.npz: outputs of the REFERENCE'S OWN tracking kernels (Core/Cuda/reduce.cu, cudafuncs.cu),
compiled from /root/reference by oracle/ref_shim/build_ref.py (g++ + a CPU SIMT emulator) and executed here, on the seeded inputs
of tests/refpin.py.  The inputs are stored next to the outputs, so the fixture is self-contained and travels to the GPU box, where
/root/reference does not exist.  Run from the repo root (needs /root/reference):  python tests/golden/make_ref_golden.py"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore", category=RuntimeWarning)

if __name__ == "__main__":
    import orc
    import ref
    import refpin
    assert ref.available(), "needs /root/reference (oracle/ref_shim/build_ref.py)"
    inp = refpin.inputs()
    out = refpin.run(ref, inp, orc.Cam)
    blob = {"in_" + k: v for k, v in inp.items()}
    blob.update({"ref_" + k: np.asarray(v) for k, v in out.items()})
    path = os.path.join(HERE, "ref_v1.npz")
    np.savez_compressed(path, **blob)
    print(path, os.path.getsize(path), "bytes,", len(out), "reference outputs")
