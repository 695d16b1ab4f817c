"""Golden trajectories of the BASELINE.json configurations at THEIR sizes, produced by the CPU oracle (test infrastructure).

The oracle needs seconds per 640x480 frame, so the long free runs are computed once here (CPU container, minutes) and
committed as small digests; tests/test_configs_gpu.py replays the same seeded streams through the HIP path on the GPU box and
compares every frame (model list, ids, surfel counts, poses bit for bit, sha256 of the label mask and of every surfel buffer).
tests/test_cpu_oracle.py re-derives the first frames of a scenario from the oracle to show the file is what the oracle produces.

    python tests/golden/make_config_golden.py [scenario ...]      ->  tests/golden/configs_v1.npz
"""
from __future__ import annotations

import hashlib
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

MAXM = 24
OUT = os.path.join(HERE, "configs_v1.npz")

# name -> parameters of the seeded stream and of the pipeline (the same keyword names tests/test_configs_gpu.py hands to the facade)
SCENARIOS = {
    # configs[2] at its own size: 4 moving objects + background, motion CRF, model spawning and deactivation
    "objects4_640": dict(W=640, H=480, n_obj=4, frames=64, multi=True, conf_global=0.5, spawn_offset=2),
    # configs[1] at its own size, long free run
    "static_640": dict(W=640, H=480, n_obj=0, frames=100, multi=False, conf_global=10.0),
    # "C2 / C3 at 300 frames" -- the same two streams played for 300 frames
    "static_640_300": dict(W=640, H=480, n_obj=0, frames=300, multi=False, conf_global=10.0),
    "objects4_640_300": dict(W=640, H=480, n_obj=4, frames=300, multi=True, conf_global=0.5, spawn_offset=2),
    # configs[0] / SURVEY's C1: `-static` on a scene whose 4 objects move unsegmented (one model, all-zero mask), 300 frames
    "static_moving4_640_300": dict(W=640, H=480, n_obj=4, frames=300, multi=False, conf_global=10.0),
    # configs[4]'s frame size (static part: one 1280x960 model)
    "static_1280": dict(W=1280, H=960, n_obj=0, frames=3, multi=False, conf_global=10.0),
    # configs[3] at its own size: 8 moving objects + background, motion CRF on (no ground-truth masks), 64 frames
    "objects8_640": dict(W=640, H=480, n_obj=8, frames=64, multi=True, conf_global=0.5, spawn_offset=2),
    # ... and for 200 frames, long enough for the CRF to have found most of the eight objects
    "objects8_640_200": dict(W=640, H=480, n_obj=8, frames=200, multi=True, conf_global=0.5, spawn_offset=2),
    # configs[4] at its own size: 1280x960, 4 moving objects + background, motion CRF on (the facade runs it with 32 M surfels per model)
    "objects4_1280": dict(W=1280, H=960, n_obj=4, frames=48, multi=True, conf_global=0.5, spawn_offset=2),
    # MORE than 16 models at a time (round 4: the cap follows max_models up to the reference's 255 ids, CoFusion.cpp:631-634): 28 moving
    # objects, ground-truth label masks (label * 9), a spawn per frame -- 27 models spawned in 34 frames, up to 19 alive together
    "objects28_320_gt": dict(W=320, H=240, n_obj=28, frames=34, multi=True, conf_global=0.5, spawn_offset=1, gt_scale=9, max_models=64),
}


def canon(a):
    """bytes of an array with every NaN replaced by the canonical quiet NaN (payloads are not part of the contract)"""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a.copy()
        a[np.isnan(a)] = np.nan
    return a.tobytes()


def digest(a):
    return hashlib.sha256(canon(a)).hexdigest()


def run_oracle(sc_name, n_frames=None, log=None):
    """yields per frame: dict(ids, counts, poses, conf, mask_sha, surf_sha)"""
    import orc_multi as om
    import orc_pipeline as op
    from co_fusion_amd import synth
    p = SCENARIOS[sc_name]
    cam = synth.Camera.scaled(p["W"], p["H"])
    sc = synth.Scene(n_obj=p["n_obj"])
    ref = (om.MultiPipeline(cam, conf_global=p["conf_global"], spawn_offset=p["spawn_offset"], max_models=p.get("max_models", 16)) if p["multi"]
           else op.StaticPipeline(cam, conf_global=p["conf_global"]))
    t0 = time.time()
    for t in range(n_frames or p["frames"]):
        d, rgb, lab, _ = sc.render(cam, t, noise=True)
        rgba = synth.rgb_to_rgba(rgb)
        if p["multi"]:
            ref.process_frame(d, rgba, gt_mask=(lab * p["gt_scale"]).astype(np.uint8) if p.get("gt_scale") else None)
            ms = ref.models
            rec = dict(ids=[m.id for m in ms], counts=[m.surfels.shape[0] for m in ms], poses=[m.pose.copy() for m in ms],
                       conf=[np.float32(m.conf_threshold) for m in ms], mask_sha=digest(ref.mask), surf_sha=[digest(m.surfels) for m in ms])
        else:
            pose, n = ref.process_frame(d, rgba)
            rec = dict(ids=[0], counts=[n], poses=[pose.copy()], conf=[np.float32(p["conf_global"])], mask_sha=digest(ref.mask),
                       surf_sha=[digest(ref.surfels)])
        if log:
            log(f"{sc_name} frame {t}: {time.time() - t0:.1f} s, models {list(zip(rec['ids'], rec['counts']))}")
        yield rec


def pack(recs):
    F = len(recs)
    out = dict(nm=np.zeros(F, np.int32), ids=np.full((F, MAXM), -1, np.int32), counts=np.zeros((F, MAXM), np.int64),
               poses=np.zeros((F, MAXM, 4, 4), np.float32), conf=np.zeros((F, MAXM), np.float32),
               mask_sha=np.array([r["mask_sha"] for r in recs]), surf_sha=np.full((F, MAXM), "", dtype="<U64"))
    for t, r in enumerate(recs):
        n = len(r["ids"])
        assert n <= MAXM
        out["nm"][t] = n
        out["ids"][t, :n] = r["ids"]; out["counts"][t, :n] = r["counts"]; out["conf"][t, :n] = r["conf"]
        for i in range(n):
            out["poses"][t, i] = r["poses"][i]
            out["surf_sha"][t, i] = r["surf_sha"][i]
    return out


def main():
    names = sys.argv[1:] or list(SCENARIOS)
    data = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        recs = list(run_oracle(name, log=lambda s: print(s, flush=True)))
        for k, v in pack(recs).items():
            data[f"{name}/{k}"] = v
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
