"""Generates tests/golden/oracle_v1.npz from the CPU oracle on seeded synthetic inputs.

These fixtures pin the ORACLE against itself (regression protection for the checker, incl. the parts that stay unpinned:
segmentation and the host GN loop).  The pins against the reference's own kernels / shaders are ref_v1.npz and
ref_surfel_v1.npz (make_ref_golden.py, make_ref_surfel_golden.py).  Large arrays are
stored as sha256 digests, small ones verbatim.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore", category=RuntimeWarning)


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def compute():
    import common
    import orc
    import orc_multi as om
    import orc_pipeline as op
    from co_fusion_amd import synth
    out = {}
    W, H = 160, 120
    cam = synth.Camera.scaled(W, H)
    ocam = orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy)
    # --- tracking: one getIncrementalTransformation on a seeded frame pair
    fp = common.frame_pair(W, H, noise=True)
    od = orc.Odometry(W, H, cam.cx, cam.cy, cam.fx, cam.fy)
    pose = common.perturbed_pose(2)
    od.init_first_rgb(fp["rgba0"]); od.init_icp_model(fp["v4"], fp["n4"], pose); od.init_rgb_model(fp["img"])
    od.init_icp(orc.depth_pyramid(fp["d1"]), 20.0); od.init_rgb(fp["rgba1"])
    vc, nc, vp, npv = (od.buffer(k, 0) for k in range(4))
    sums, err = orc.icp_step(pose[:3, :3], pose[:3, 3], vc, nc, np.linalg.inv(pose[:3, :3].astype(np.float64)).astype(np.float32),
                             pose[:3, 3], ocam, vp, npv, 0.10, np.float32(np.sin(20.0 * 3.14159254 / 180.0)), want_err=True)
    out["icp_sums"] = sums
    out["icp_err_sha"] = digest(err)
    tr, rot, st = od.track(pose[:3, 3], pose[:3, :3])
    out["track_trans"] = tr; out["track_rot"] = rot
    out["track_counts"] = np.array([st.last_icp_count, st.last_rgb_count, st.so3_iterations], np.float64)
    # --- static pipeline: poses + surfel counts + buffer digest over 5 frames
    sc = synth.Scene(n_obj=1)
    pl = op.StaticPipeline(cam, conf_global=0.5)
    poses, counts = [], []
    for t in range(5):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        p, n = pl.process_frame(d, synth.rgb_to_rgba(rgb))
        poses.append(p); counts.append(n)
    out["static_poses"] = np.stack(poses); out["static_counts"] = np.array(counts, np.int64)
    out["static_surfels_sha"] = digest(pl.surfels)
    # --- segmentation: SLIC digest + CRF marginals on random unaries
    d, rgb, _, _ = sc.render(cam, 3, noise=True)
    lab = om.slic(synth.rgb_to_rgba(rgb))
    out["slic_sha"] = digest(lab)
    out["slic_counts"] = np.bincount(lab.ravel(), minlength=(W // 16) * (H // 16)).astype(np.int64)
    K = (W // 16) * (H // 16); L = 3
    rng = np.random.default_rng(5)
    unary = rng.uniform(0, 5, size=(K, L)).astype(np.float32)
    f1 = np.stack([(np.arange(K) % (W // 16)) / 2.0, (np.arange(K) // (W // 16)) / 2.0], -1).astype(np.float32)
    f2 = rng.uniform(0, 6, size=(K, 6)).astype(np.float32)
    Q = np.zeros((K, L), np.float32)
    import ctypes as C
    orc.lib.orc_crf_meanfield(orc.P(unary), L, K, orc.P(f1), orc.P(f2), C.c_float(2.0), C.c_float(7.0), 10, orc.P(Q))
    out["crf_Q"] = Q
    # --- multi-model pipeline with GT masks: model counts and surfel counts
    sc2 = synth.Scene(n_obj=2)
    mp = om.MultiPipeline(cam, conf_global=0.5, spawn_offset=2)
    hist = []
    for t in range(6):
        d, rgb, lab, _ = sc2.render(cam, t, noise=True)
        mp.process_frame(d, synth.rgb_to_rgba(rgb), gt_mask=(lab * 40).astype(np.uint8))
        hist.append([len(mp.models)] + [m.surfels.shape[0] for m in mp.models] + [0] * (4 - len(mp.models)))
    out["multi_hist"] = np.array(hist, np.int64)
    out["multi_mask_sha"] = digest(mp.mask)
    return out


if __name__ == "__main__":
    o = compute()
    np.savez_compressed(os.path.join(HERE, "oracle_v1.npz"), **o)
    print({k: (v.shape, v.dtype) for k, v in o.items()})
