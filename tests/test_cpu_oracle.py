"""CPU suite: the oracle against its golden fixtures and against independent properties.  No GPU needed."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import common
import orc
import orc_multi as om
import orc_pipeline as op
from co_fusion_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_golden_fixtures_match():
    """tests/golden/oracle_v1.npz (made by tests/golden/make_golden.py) pins the oracle against itself."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    ref = np.load(os.path.join(HERE, "golden", "oracle_v1.npz"))
    cur = make_golden.compute()
    assert set(ref.files) == set(cur.keys())
    for k in ref.files:
        assert np.array_equal(ref[k], cur[k]), f"golden mismatch in {k}"


def _tracker(fp, W, H, cam, pose):
    od = orc.Odometry(W, H, cam.cx, cam.cy, cam.fx, cam.fy)
    od.init_first_rgb(fp["rgba0"]); od.init_icp_model(fp["v4"], fp["n4"], pose); od.init_rgb_model(fp["img"])
    od.init_icp(orc.depth_pyramid(fp["d1"]), 20.0); od.init_rgb(fp["rgba1"])
    return od


def test_fixed_point_sums_agree_with_reference_order_f32_tree():
    """The exact fixed-point statement of icpStep vs the reference's own f32 grid-stride + warp-tree order
    (reduce.cu:90-185) for two launch shapes: relative spread <= 1e-4, inliers identical."""
    W, H = 160, 120
    fp = common.frame_pair(W, H)
    cam = fp["cam"]
    pose = common.perturbed_pose(2)
    od = _tracker(fp, W, H, cam, pose)
    vc, nc, vp, npv = (od.buffer(k, 0) for k in range(4))
    Rinv = np.linalg.inv(pose[:3, :3].astype(np.float64)).astype(np.float32)
    angle = np.float32(np.sin(20.0 * 3.14159254 / 180.0))
    ocam = orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy)
    sums, _ = orc.icp_step(pose[:3, :3], pose[:3, 3], vc, nc, Rinv, pose[:3, 3], ocam, vp, npv, 0.10, angle)
    exact = sums[:28].astype(np.float64) / 2.0 ** 32
    assert sums[28] > 0.5 * W * H
    scale = np.abs(exact).max()
    for threads, blocks in [(128, 112), (256, 96)]:  # GPUConfig.h:51-52 default and the TITAN X entry :71
        tree = orc.icp_step_f32tree(pose[:3, :3], pose[:3, 3], vc, nc, Rinv, pose[:3, 3], ocam, vp, npv, 0.10, angle, threads, blocks)
        assert tree[28] == sums[28]
        assert np.abs(tree[:28] - exact).max() / scale < 1e-4


def test_tracker_recovers_known_motion_on_clean_data():
    W, H = 320, 240
    fp = common.frame_pair(W, H, noise=False, t0=0, t1=2)
    cam = fp["cam"]
    pose = np.eye(4, dtype=np.float32)
    od = _tracker(fp, W, H, cam, pose)
    tr, rot, st = od.track(pose[:3, 3], pose[:3, :3])
    T1 = fp["T1"]
    assert np.linalg.norm(tr - T1[:3, 3]) < 3e-3
    assert np.abs(rot - T1[:3, :3]).max() < 3e-3
    assert st.last_icp_count > 0.8 * W * H


def test_deterministic_math_against_libm():
    lib = orc.lib
    # orc_expf / orc_acosf are static inline; exercise them through the bilateral filter and the CRF instead:
    # a constant-depth image must be a fixed point of the bilateral filter
    d = np.full((48, 64), 1.25, np.float32)
    out = op.bilateral(d, 5.0)
    assert np.allclose(out, 1.25, rtol=0, atol=2e-6)  # 169 f32 taps: summation rounding only
    # softmax rows of the mean field sum to one and favour the low-unary label
    K, L = 70, 3
    unary = np.tile(np.array([0.1, 3.0, 3.0], np.float32), (K, 1))
    f1 = np.stack([(np.arange(K) % 10) / 2.0, (np.arange(K) // 10) / 2.0], -1).astype(np.float32)
    f2 = np.zeros((K, 6), np.float32)
    Q = np.zeros((K, L), np.float32)
    lib.orc_crf_meanfield(orc.P(unary), L, K, orc.P(f1), orc.P(f2), C.c_float(2.0), C.c_float(7.0), 5, orc.P(Q))
    assert np.allclose(Q.sum(1), 1, atol=1e-6) and (Q.argmax(1) == 0).all()


def test_fusion_weight_limits():
    I = np.eye(4, dtype=np.float32)
    assert op.fusion_weight(I, I, 1.0) == 1.0
    assert op.fusion_weight(I, I, 100.0) == 100.0
    far = I.copy(); far[0, 3] = 0.05
    assert op.fusion_weight(far, I, 1.0) == 0.5          # clipped at `largest`, floor minWeight (Model.cpp:399-404)
    near = I.copy(); near[2, 3] = 0.002
    assert abs(op.fusion_weight(near, I, 1.0) - 0.8) < 1e-4


def test_index_map_nearest_surfel_wins_and_first_on_ties():
    cam = orc.Cam(100, 100, 32, 24)
    s = np.zeros((4, 12), np.float32)
    s[:, 2] = [1.0, 0.5, 0.5, 2.0]         # all on the optical axis -> same pixel (32, 24)
    s[:, 3] = 1; s[:, 8:11] = [0, 0, 1]; s[:, 11] = 0.01
    idx, vc, ct, nr = op.predict_indices(s, np.eye(4, dtype=np.float32), cam, 64, 48, 20.0, 1, op.StaticPipeline.TIME_DELTA)
    assert idx[24, 32] == 1 and vc[24, 32, 2] == 0.5     # nearest z, and the FIRST of the two equal ones
    assert (idx > 0).sum() == 1


def test_clean_is_order_preserving_and_drops_stale_unstable():
    cam = orc.Cam(100, 100, 32, 24)
    s = np.zeros((3, 12), np.float32)
    s[:, 2] = [1.0, 1.1, 1.2]; s[:, 0] = [0.0, 0.05, -0.05]
    s[:, 3] = [5.0, 0.1, 5.0]; s[:, 6] = 1; s[:, 7] = [30, 1, 30]   # the middle one: old and unconfident -> dropped
    s[:, 8:11] = [0, 0, 1]; s[:, 11] = 0.01
    pose = np.eye(4, dtype=np.float32)
    idx, vc, ct, nr = op.predict_indices(s, pose, cam, 64, 48, 20.0, 30, op.StaticPipeline.TIME_DELTA)
    depth = np.zeros((48, 64), np.float32); mask = np.zeros((48, 64), np.uint8)
    out = op.clean(s, np.zeros((0, 12), np.float32), idx, vc, ct, depth, mask, pose, cam, 30, 1.0, 3.0, op.StaticPipeline.TIME_DELTA, 0)
    assert out.shape[0] == 2 and np.array_equal(out[:, 2], np.array([1.0, 1.2], np.float32))


def test_static_pipeline_counts_are_stable_and_tracking_follows():
    W, H = 160, 120
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=0)
    pl = op.StaticPipeline(cam, conf_global=0.5)
    for t in range(4):
        d, rgb, _, T = sc.render(cam, t, noise=False)
        pose, n = pl.process_frame(d, synth.rgb_to_rgba(rgb))
        assert 0.8 * W * H < n < 1.5 * W * H
    assert np.linalg.norm(pose[:3, 3] - T[:3, 3]) < 1e-2


def test_gt_mask_segmentation_branch():
    d = np.full((32, 48), 2.0, np.float32); d[:, 24:] = 1.0
    gt = np.zeros((32, 48), np.uint8); gt[:, 24:] = 80
    mapping = np.zeros(256, np.uint8)
    r = om.segment_gt(gt, d, [0], 1, True, mapping)
    assert r["hasNewLabel"] and (r["full"][:, 24:] == 1).all() and (r["full"][:, :24] == 0).all()
    assert mapping[80] == 1
    assert abs(r["modelData"][0]["depthMean"] - 2.0) < 1e-6 and abs(r["modelData"][1]["depthMean"] - 1.0) < 1e-6
    r2 = om.segment_gt(gt, d, [0, 1], 2, True, mapping)   # label already mapped: no second spawn
    assert not r2["hasNewLabel"] and len(r2["modelData"]) == 2


def test_rgb_only_tracking_moves_towards_the_known_motion():
    """rgbOnly hands sigma = -1 (unit weights) to rgbStep: rows are ~count times larger than in the weighted mode.  With the
    sigma-dependent fixed-point window the photometric-only step is finite and reduces the pose error (it wrapped before)."""
    W, H = 320, 240
    fp = common.frame_pair(W, H, noise=False, t0=0, t1=2)
    pose = np.eye(4, dtype=np.float32)
    od = _tracker(fp, W, H, fp["cam"], pose)
    tr, rot, st = od.track(pose[:3, 3], pose[:3, :3], rgb_only=True)
    T1 = fp["T1"]
    assert np.isfinite(tr).all() and np.isfinite(rot).all() and st.last_rgb_count > 5000
    assert np.linalg.norm(tr - T1[:3, 3]) < np.linalg.norm(T1[:3, 3])
    assert np.abs(rot - T1[:3, :3]).max() < np.abs(np.eye(3) - T1[:3, :3]).max()


def test_config_golden_head_is_what_the_oracle_produces():
    """tests/golden/configs_v1.npz (the oracle's free runs at BASELINE.json's sizes, consumed by tests/test_configs_gpu.py on the GPU box)
    is reproducible: the first frames of the 640x480 static scenario and of the 4-object scenario, re-derived here."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_config_golden as mcg
    g = np.load(os.path.join(here, "golden", "configs_v1.npz"))
    for name, frames in (("static_640", 2), ("objects4_640", 2)):
        for t, rec in enumerate(mcg.run_oracle(name, n_frames=frames)):
            n = int(g[f"{name}/nm"][t])
            assert n == len(rec["ids"])
            assert list(g[f"{name}/ids"][t, :n]) == rec["ids"] and list(g[f"{name}/counts"][t, :n]) == rec["counts"]
            for i in range(n):
                assert g[f"{name}/poses"][t, i].tobytes() == np.asarray(rec["poses"][i], np.float32).tobytes(), f"{name} frame {t}: pose"
                assert str(g[f"{name}/surf_sha"][t, i]) == rec["surf_sha"][i], f"{name} frame {t}: surfel buffer"
            assert str(g[f"{name}/mask_sha"][t]) == rec["mask_sha"]
