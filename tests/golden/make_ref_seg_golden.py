"""tests/golden/ref_seg_v1.npz: what the REFERENCE'S OWN Core/Segmentation sources (Segmentation.cpp, Slic.h, Slic.cpp,
ConnectedLabels.hpp -- compiled from /root/reference by oracle/ref_shim/build_ref.py; gSLICr / densecrf stand-ins delegate to the
oracle's SLIC and exact mean-field operations) answer for every performSegmentationCRF call of the tests/segpin.py scenario.
Inputs are regenerated at test time from the seeded oracle pipeline and checked against the stored digests.

    python tests/golden/make_ref_seg_golden.py        (needs /root/reference; ~15 s)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import segpin  # noqa: E402


def main():
    calls = segpin.capture()
    out = dict(n_calls=np.array([len(calls)]))
    for i, c in enumerate(calls):
        out[f"c{i}/input_sha"] = np.array(segpin.input_digest(c))
        for k, v in segpin.pack_result(segpin.run_reference(c)).items():
            out[f"c{i}/{k}"] = np.array(v)
    path = os.path.join(HERE, "ref_seg_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(calls), "calls; models per call:", [len(c["ids"]) for c in calls])


if __name__ == "__main__":
    main()
