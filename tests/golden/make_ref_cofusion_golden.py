#!/usr/bin/env python3
"""tests/golden/ref_cofusion_v1.json: per-frame digests of what the REFERENCE'S OWN frame loop decides -- CoFusion::processFrame and its
helpers, cut out of /root/reference/Core/CoFusion.cpp at build time and compiled behind oracle/ref_shim/stub/CoFusionPin.h, with the
reference's own Core/Segmentation and the CPU oracle's passes -- for the scenarios of tests/cfpin.py (spawning, deactivation, id
re-use, confidence thresholds, ground-truth masks, fill-in tracking, the single-model mode).  Needs /root/reference; ~30 s.
Every scenario runs in a process of its own (see cfpin.run_reference_isolated for why)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import cfpin  # noqa: E402


def main():
    out = {"size": [cfpin.W, cfpin.H], "scenarios": {}}
    for name in cfpin.SCENARIOS:
        rows = cfpin.run_reference_isolated(name)
        out["scenarios"][name] = rows
        print(name, [len(r["ids"]) for r in rows], flush=True)
    with open(os.path.join(HERE, "ref_cofusion_v1.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
