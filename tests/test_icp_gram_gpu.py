"""GPU parity of the Gram form of the ICP sums (cf_set_icp_arith 1: row entries rounded once to a fixed-point grid, the 29 sums
contracted over the pixels by v_mfma_i32_32x32x32_i8 on signed 8-bit limbs) against the oracle under the same rounding specification
(oracle/orc.h: ORC_ICP_ARITH_GRAM): bit-exact per launch, per Gauss-Newton schedule and free running through the facade -- the same
bars as the product form has in test_track_gpu.py / test_facade_gpu.py."""
import warnings

import numpy as np
import pytest

import common
import orc
import orc_multi as om
import orc_pipeline as op
from co_fusion_amd import synth

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", category=RuntimeWarning)


@pytest.fixture(autouse=True)
def gram_oracle():
    orc.set_icp_arith("gram")
    yield
    orc.set_icp_arith("product")


@pytest.fixture(scope="module")
def ctx():
    from co_fusion_amd import api
    c = api.Context(640, 480, 528, 528, 320, 240)
    c.set_icp_arith("gram")
    yield c
    c.close()


def _eq(a, b, what):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ok = (a == b) | ((a != a) & (b != b)) if a.dtype.kind == "f" else (a == b)
    assert ok.all(), f"{what}: {np.count_nonzero(~ok)} of {ok.size} values differ (first at {np.argwhere(~ok)[0]}: {a[~ok][0]} vs {b[~ok][0]})"


def _oracle_tracker(fp, W=640, H=480, cam=None, seed=3, v4=None, n4=None):
    cam = cam or orc.Cam(528, 528, 320, 240)
    od = orc.Odometry(W, H, cam.cx, cam.cy, cam.fx, cam.fy)
    od.init_first_rgb(fp["rgba0"])
    pose = common.perturbed_pose(seed)
    od.init_icp_model(fp["v4"] if v4 is None else v4, fp["n4"] if n4 is None else n4, pose)
    od.init_rgb_model(fp["img"])
    od.init_icp(orc.depth_pyramid(fp["d1"]), 20.0)
    od.init_rgb(fp["rgba1"])
    return od, pose


@pytest.mark.parametrize("threads,ppt", [(256, 1), (256, 2), (256, 4), (1024, 1), (1024, 4), (64, 4), (64, 1)])
def test_icp_step_gram_exact(ctx, threads, ppt):
    """one icpStep launch at the three pyramid levels, every launch shape: the 29 integer sums, A / b / residual and the error surface"""
    from co_fusion_amd import api
    fp = common.frame_pair()
    od, pose = _oracle_tracker(fp)
    ctx.set_icp_launch(threads, ppt)
    Rprev = pose[:3, :3]; tprev = pose[:3, 3]
    Rprev_inv = np.linalg.inv(Rprev.astype(np.float64)).astype(np.float32)
    T2 = common.perturbed_pose(7, 0.004, 0.3) @ pose
    angle = np.float32(np.sin(20.0 * 3.14159254 / 180.0))
    try:
        for lvl in range(3):
            vc, nc, vp, npv = (od.buffer(k, lvl) for k in range(4))
            cam_l = orc.Cam(528, 528, 320, 240).level(lvl)
            osums, oerr = orc.icp_step(T2[:3, :3], T2[:3, 3], vc, nc, Rprev_inv, tprev, cam_l, vp, npv, 0.10, angle, want_err=True)
            err = ctx.empty(oerr.shape)
            A, b, res, sums = ctx.icp_step(T2[:3, :3], T2[:3, 3], ctx.to_device(vc), ctx.to_device(nc), Rprev_inv, tprev,
                                           api.Cam(528, 528, 320, 240).level(lvl), ctx.to_device(vp), ctx.to_device(npv), 0.10, angle, err_surface=err)
            assert osums[28] > 0.5 * vc.shape[1] * vc.shape[0] / 3, "test scene should have plenty of inliers"
            _eq(sums[:29], osums[:29], f"Gram sums L{lvl}")
            oA, ob, ores = orc.icp_sums_to_host(osums)
            _eq(A, oA, "A"); _eq(b, ob, "b"); _eq(res, ores, "residual")
            _eq(err.cpu().numpy(), oerr, "ICP error surface")
            # the two rounding specifications describe the same normal equations: entries agree to ~1e-6 of the largest one
            orc.set_icp_arith("product")
            psums, _ = orc.icp_step(T2[:3, :3], T2[:3, 3], vc, nc, Rprev_inv, tprev, cam_l, vp, npv, 0.10, angle)
            pA, pb, pres = orc.icp_sums_to_host(psums)
            orc.set_icp_arith("gram")
            assert pres[1] == res[1]
            assert np.abs(pA - A).max() <= 2e-6 * np.abs(pA).max() and np.abs(pb - b).max() <= 2e-6 * max(np.abs(pb).max(), 1.0), (np.abs(pA - A).max(), np.abs(pb - b).max())
            # a row band (one rank's share of a split model): bands add up to the whole image exactly
            H_l = vc.shape[0] // 3
            parts = [ctx.icp_step_band(T2[:3, :3], T2[:3, 3], ctx.to_device(vc), ctx.to_device(nc), Rprev_inv, tprev, api.Cam(528, 528, 320, 240).level(lvl),
                                       ctx.to_device(vp), ctx.to_device(npv), 0.10, angle, r0, r1) for r0, r1 in ((0, H_l // 3), (H_l // 3, H_l))]
            _eq((parts[0] + parts[1])[:29], osums[:29], f"Gram sums L{lvl}, two row bands")
    finally:
        ctx.set_icp_launch(256, 1)


@pytest.mark.parametrize("opts", [dict(), dict(so3=False), dict(pyramid=False), dict(fast_odom=True), dict(icp_weight=100.0)])
def test_gauss_newton_loop_gram(ctx, opts):
    """the whole device-resident schedule (SO(3) + 4/5/10 iterations) against the oracle's host loop, both on the Gram form: pose bits"""
    from co_fusion_amd import api
    fp = common.frame_pair(noise=True)
    od, pose = _oracle_tracker(fp, seed=2)
    g = api.Odometry(ctx)
    d = ctx.to_device
    g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(fp["v4"]), d(fp["n4"]), pose); g.init_rgb_model(d(fp["img"]))
    g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
    oerr = np.zeros((480, 640), np.float32)
    otr, orot, ost = od.track(pose[:3, 3], pose[:3, :3], err_surface=oerr, **opts)
    err = ctx.empty((480, 640)); err.zero_()
    tr, rot, st = g.track(pose[:3, 3], pose[:3, :3], err_surface=err, **opts)
    assert np.asarray(tr, np.float32).tobytes() == np.asarray(otr, np.float32).tobytes(), (tr, otr)
    assert np.asarray(rot, np.float32).tobytes() == np.asarray(orot, np.float32).tobytes()
    assert st.last_icp_count == ost.last_icp_count and st.last_rgb_count == ost.last_rgb_count
    np.testing.assert_allclose(st.last_icp_error, ost.last_icp_error, rtol=1e-6)
    _eq(err.cpu().numpy(), oerr, "ICP error surface after tracking")
    _eq(np.array(st.lastA), np.array(ost.lastA), "last normal equations")
    # ... and close to the product form's pose: the two specifications differ by rounding noise (1e-6 m; with the ICP weight at 100 the
    # photometric term no longer conditions the system and the noise is amplified fifty-fold -- as it is for the reference's f32 tree)
    orc.set_icp_arith("product")
    od2, _ = _oracle_tracker(fp, seed=2)
    ptr, prot, _ = od2.track(pose[:3, 3], pose[:3, :3], **opts)
    orc.set_icp_arith("gram")
    bound = 2e-4 if "icp_weight" in opts else 1e-6
    assert np.abs(np.asarray(ptr) - np.asarray(tr)).max() <= bound and np.abs(np.asarray(prot) - np.asarray(rot)).max() <= bound
    g.close()


def test_culled_object_model_gram():
    """an object-sized model with the occupancy look-up and the screen-box culling: culled = unculled = oracle, on the Gram form"""
    from co_fusion_amd import api
    W, H = 320, 240
    fp = common.frame_pair(W, H, noise=True)
    cam = fp["cam"]
    pose = common.perturbed_pose(2)
    v4 = fp["v4"].copy(); n4 = fp["n4"].copy()
    keep = np.zeros((H, W), bool); keep[60:150, 120:200] = True
    v4[~keep] = 0; n4[~keep] = 0
    c = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
    c.set_icp_arith("gram")
    d = c.to_device

    def run(cull):
        g = api.Odometry(c)
        g.set_culling(cull)
        g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(v4), d(n4), pose); g.init_rgb_model(d(fp["img"]))
        g.init_icp(c.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
        tr, rot, st = g.track(pose[:3, 3], pose[:3, :3])
        g.close()
        return np.asarray(tr, np.float32), np.asarray(rot, np.float32), st

    t0, r0, s0 = run(False)
    t1, r1, s1 = run(True)
    od, _ = _oracle_tracker(fp, W, H, orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy), seed=2, v4=v4, n4=n4)
    otr, orot, ost = od.track(pose[:3, 3], pose[:3, :3])
    assert t0.tobytes() == t1.tobytes() == np.asarray(otr, np.float32).tobytes()
    assert r0.tobytes() == r1.tobytes() == np.asarray(orot, np.float32).tobytes()
    assert s0.last_icp_count == s1.last_icp_count == ost.last_icp_count > 100
    c.close()


def test_facade_free_run_gram():
    """the C++ facade free running on the Gram form against the oracle's frame loop on the Gram form: a static scene for 6 frames, two
    moving objects with the motion CRF (spawns) for 8 frames -- poses, counts, surfel buffers, label masks bit for bit"""
    from co_fusion_amd import facade
    W, H = 320, 240
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=0)
    ref = op.StaticPipeline(cam, conf_global=0.5)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, enable_multiple_models=0)
    cf.set_icp_arith("gram")
    for t in range(6):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        rp, rn = ref.process_frame(d, synth.rgb_to_rgba(rgb))
        cf.process_frame(d, rgb, timestamp=t)
        info = cf.model_info(0)
        assert info["count"] == rn, f"frame {t}: count"
        _eq(info["pose"], rp, f"frame {t}: pose")
        _eq(cf.model_download(0), ref.surfels, f"frame {t}: surfels")
    cf.close()
    sc = synth.Scene(n_obj=2)
    ref = om.MultiPipeline(cam, conf_global=0.5, spawn_offset=2)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, model_spawn_offset=2, enable_multiple_models=1)
    cf.set_icp_arith("gram")
    spawned = False
    for t in range(8):
        d, rgb, lab, _ = sc.render(cam, t, noise=True)
        ref.process_frame(d, synth.rgb_to_rgba(rgb))
        cf.process_frame(d, rgb, timestamp=t)
        assert cf.num_models == len(ref.models), f"frame {t}: model count {cf.num_models} vs {len(ref.models)}"
        if t > 0:
            _eq(cf.mask(), ref.mask, f"frame {t}: label mask")
        for i, m in enumerate(ref.models):
            info = cf.model_info(i)
            assert info["id"] == m.id and info["count"] == m.surfels.shape[0], f"frame {t} model {i}"
            _eq(info["pose"], m.pose, f"frame {t} model {i}: pose")
            _eq(cf.model_download(i), m.surfels, f"frame {t} model {i}: surfels")
        spawned = spawned or len(ref.models) > 1
    assert spawned, "no object model was spawned: the multi-model path was not exercised"
    cf.close()
