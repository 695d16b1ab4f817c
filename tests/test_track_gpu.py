"""GPU parity of the tracking half of the hot path: HIP (through the C-ABI) vs the CPU oracle.

Integer/index outputs (validity, correspondences, counts, fixed-point sums) must be bit-exact;
per-pixel f32 outputs are bit-exact too because both sides use the same IEEE operation order.
"""
import numpy as np
import pytest

import common
import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from co_fusion_amd import api
    c = api.Context(640, 480, 528, 528, 320, 240)
    yield c
    c.close()


def _eq(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, what
    same = (a.view(np.uint8) == b.view(np.uint8)) if a.dtype.kind == "V" else ((a == b) | ((a != a) & (b != b)))
    assert same.all(), f"{what}: {np.count_nonzero(~same)} of {same.size} differ"


def _planar_valid_only(m):
    """y/z planes are undefined where x is NaN (reference writes the x plane only)."""
    m = m.copy()
    h = m.shape[0] // 3
    bad = np.isnan(m[:h])
    m[h:2 * h][bad] = 0; m[2 * h:][bad] = 0
    return m


def test_map_prep_parity(ctx):
    from co_fusion_amd import api
    fp = common.frame_pair(noise=True)
    cam = api.Cam(528, 528, 320, 240); ocam = orc.Cam(528, 528, 320, 240)
    d = ctx.to_device(fp["d1"])
    for lvl, dep in enumerate(orc.depth_pyramid(fp["d1"])):
        dd = ctx.to_device(dep)
        v = ctx.create_vmap(dd, cam.level(lvl), 20.0)
        ov = orc.create_vmap(dep, ocam.level(lvl), 20.0)
        _eq(_planar_valid_only(v.cpu().numpy()), _planar_valid_only(ov), f"vmap L{lvl}")
        n = ctx.create_nmap(v)
        on = orc.create_nmap(ov)
        _eq(_planar_valid_only(n.cpu().numpy()), _planar_valid_only(on), f"nmap L{lvl}")
        if lvl < 2:
            _eq(_planar_valid_only(ctx.resize_map(v, False).cpu().numpy()), _planar_valid_only(orc.resize_map(ov, False)), "resize v")
            _eq(_planar_valid_only(ctx.resize_map(n, True).cpu().numpy()), _planar_valid_only(orc.resize_map(on, True)), "resize n")
    pyr = ctx.depth_pyramid(d)
    opyr = orc.depth_pyramid(fp["d1"])
    _eq(pyr[1].cpu().numpy(), opyr[1], "depth pyr 1")
    _eq(pyr[2].cpu().numpy(), opyr[2], "depth pyr 2")
    v4 = ctx.to_device(fp["v4"]); n4 = ctx.to_device(fp["n4"])
    v, n = ctx.copy_maps(v4, n4)
    ov, on = orc.copy_maps(fp["v4"], fp["n4"])
    _eq(v.cpu().numpy(), ov, "copy_maps v"); _eq(n.cpu().numpy(), on, "copy_maps n")
    T = common.perturbed_pose(1)
    ctx.transform_maps(v, n, T[:3, :3], T[:3, 3])
    tv, tn = orc.transform_maps(ov, on, T[:3, :3], T[:3, 3])
    _eq(v.cpu().numpy(), tv, "transform v"); _eq(n.cpu().numpy(), tn, "transform n")
    _eq(ctx.vertices_to_depth(v4, 6.0).cpu().numpy(), orc.vertices_to_depth(fp["v4"], 6.0), "vertices_to_depth")
    img = ctx.rgba_to_intensity(ctx.to_device(fp["rgba1"]))
    oimg = orc.rgba_to_intensity(fp["rgba1"])
    _eq(img.cpu().numpy(), oimg, "intensity")
    i1 = ctx.pyrdown_gauss_u8(img); oi1 = orc.pyrdown_gauss_u8(oimg)
    _eq(i1.cpu().numpy(), oi1, "pyrdown u8")
    dx, dy = ctx.sobel(img); odx, ody = orc.sobel(oimg)
    _eq(dx.cpu().numpy(), odx, "sobel dx"); _eq(dy.cpu().numpy(), ody, "sobel dy")
    cl = ctx.project_cloud(ctx.to_device(opyr[1]), cam.level(1))
    _eq(cl.cpu().numpy(), orc.project_cloud(opyr[1], ocam.level(1)), "cloud")


def _tracker_inputs(fp, noise_pose_seed=3):
    """Build an oracle tracker with all pyramids, return it + the perturbed start pose."""
    od = orc.Odometry(640, 480, 320, 240, 528, 528)
    od.init_first_rgb(fp["rgba0"])
    pose = common.perturbed_pose(noise_pose_seed)
    od.init_icp_model(fp["v4"], fp["n4"], pose)
    od.init_rgb_model(fp["img"])
    od.init_icp(orc.depth_pyramid(fp["d1"]), 20.0)
    od.init_rgb(fp["rgba1"])
    return od, pose


@pytest.mark.parametrize("threads,ppt", [(256, 1), (256, 2), (256, 4), (1024, 1), (64, 4)])
def test_icp_step_exact(ctx, threads, ppt):
    from co_fusion_amd import api
    fp = common.frame_pair()
    od, pose = _tracker_inputs(fp)
    ctx.set_icp_launch(threads, ppt)
    Rprev = pose[:3, :3]; tprev = pose[:3, 3]
    Rprev_inv = np.linalg.inv(Rprev.astype(np.float64)).astype(np.float32)
    T2 = common.perturbed_pose(7, 0.004, 0.3) @ pose
    angle = np.float32(np.sin(20.0 * 3.14159254 / 180.0))
    for lvl in range(3):
        vc, nc, vp, npv = (od.buffer(k, lvl) for k in range(4))
        osums, oerr = orc.icp_step(T2[:3, :3], T2[:3, 3], vc, nc, Rprev_inv, tprev, orc.Cam(528, 528, 320, 240).level(lvl),
                                   vp, npv, 0.10, angle, want_err=True)
        err = ctx.empty(oerr.shape)
        A, b, res, sums = ctx.icp_step(T2[:3, :3], T2[:3, 3], ctx.to_device(vc), ctx.to_device(nc), Rprev_inv, tprev,
                                       api.Cam(528, 528, 320, 240).level(lvl), ctx.to_device(vp), ctx.to_device(npv),
                                       0.10, angle, err_surface=err)
        assert osums[28] > 0.5 * vc.shape[1] * vc.shape[0] / 3, "test scene should have plenty of inliers"
        _eq(sums, osums, f"ICP fixed-point sums L{lvl}")
        oA, ob, ores = orc.se3_to_host(osums)
        _eq(A, oA, "A"); _eq(b, ob, "b"); _eq(res, ores, "residual")
        _eq(err.cpu().numpy(), oerr, "ICP error surface")
    ctx.set_icp_launch(256, 1)


def test_rgb_and_so3_steps_exact(ctx):
    fp = common.frame_pair()
    od, pose = _tracker_inputs(fp)
    # derivative images come from a first oracle track call; re-create inputs afterwards
    od.track(pose[:3, 3], pose[:3, :3])
    od2, _ = _tracker_inputs(fp)
    K = np.array([[528, 0, 320], [0, 528, 240], [0, 0, 1]], np.float64)
    dT = common.perturbed_pose(11, 0.003, 0.2).astype(np.float64)
    for lvl in range(3):
        Kl = K.copy(); Kl[:2] /= (1 << lvl)
        nextImage = od2.buffer(7, lvl); lastImage = od2.buffer(6, lvl)
        lastDepth = od2.buffer(4, lvl); nextDepth = od2.buffer(5, lvl)
        dx, dy = orc.sobel(nextImage)
        krkinv = (Kl @ dT[:3, :3] @ np.linalg.inv(Kl)).astype(np.float32)
        kt = (Kl @ dT[:3, 3]).astype(np.float32)
        min_scale = float([5, 3, 1][lvl] ** 2 / 0.125 ** 2)
        ocor, osig, ocnt = orc.rgb_residual(min_scale, dx, dy, lastDepth, nextDepth, lastImage, nextImage, 0.07, kt, krkinv)
        d = ctx.to_device
        ddx, ddy = d(dx), d(dy)
        cor, sig, cnt = ctx.rgb_residual(min_scale, ddx, ddy, d(lastDepth), d(nextDepth), d(lastImage), d(nextImage), 0.07,
                                         kt, krkinv)
        assert ocnt > 1000
        assert (sig, cnt) == (osig, ocnt)
        _eq(cor.cpu().numpy().view(orc.DATATERM).reshape(-1), ocor, f"DataTerm L{lvl}")
        cloud = orc.project_cloud(lastDepth, orc.Cam(528, 528, 320, 240).level(lvl))
        for sigma in (float(ocnt), 1.0, -1.0):
            osums = orc.rgb_step(ocor, sigma, cloud, 528.0 / (1 << lvl), 528.0 / (1 << lvl), dx, dy, 0.125)
            A, b, sums = ctx.rgb_step(cor, sigma, d(cloud), 528.0 / (1 << lvl), 528.0 / (1 << lvl), ddx, ddy, 0.125)
            _eq(sums[:29], osums[:29], f"RGB sums L{lvl} sigma={sigma}")
            oA, ob, _ = orc.se3_to_host(osums, orc.rgb_fix_bits(sigma))
            _eq(A, oA, "A rgb"); _eq(b, ob, "b rgb")
    # SO3 at level 2
    last2, next2 = od2.buffer(8, 2), od2.buffer(7, 2)
    K2 = K.copy(); K2[:2] /= 4
    Rr = common.perturbed_pose(5, 0.0, 0.4)[:3, :3].astype(np.float64)
    basis = (K2 @ Rr @ np.linalg.inv(K2)).astype(np.float32)
    kinv = np.linalg.inv(K2).astype(np.float32)
    krlr = (K2 @ Rr).astype(np.float32)
    osums = orc.so3_step(last2, next2, basis, kinv, krlr)
    A, b, res, sums = ctx.so3_step(ctx.to_device(last2), ctx.to_device(next2), basis, kinv, krlr)
    _eq(sums[:11], osums[:11], "SO3 sums")
    oA, ob, ores = orc.so3_to_host(osums)
    _eq(A, oA, "A so3"); _eq(b, ob, "b so3"); _eq(res, ores, "res so3")


@pytest.mark.parametrize("opts", [dict(), dict(so3=False), dict(pyramid=False), dict(fast_odom=True),
                                  dict(icp_weight=100.0), dict(rgb_only=True)])
def test_get_incremental_transformation(ctx, opts):
    """Whole device-resident GN loop vs the oracle's host loop: same pose bits expected (exact sums,
    identical f64 solve); asserted to 1e-6 so that a libm-level difference cannot flake the gate."""
    from co_fusion_amd import api
    fp = common.frame_pair(noise=True)
    od, pose = _tracker_inputs(fp, noise_pose_seed=2)
    g = api.Odometry(ctx)
    d = ctx.to_device
    g.init_first_rgb(d(fp["rgba0"]))
    g.init_icp_model(d(fp["v4"]), d(fp["n4"]), pose)
    g.init_rgb_model(d(fp["img"]))
    g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0)
    g.init_rgb(d(fp["rgba1"]))
    for which in range(9):
        for lvl in range(3):
            a, b = g.buffer(which, lvl), od.buffer(which, lvl)
            if which <= 3:
                a, b = _planar_valid_only(a), _planar_valid_only(b)
            _eq(a, b, f"pyramid buffer {which} L{lvl}")
    oerr = np.zeros((480, 640), np.float32)
    otr, orot, ost = od.track(pose[:3, 3], pose[:3, :3], err_surface=oerr, **opts)
    err = ctx.empty((480, 640)); err.zero_()
    tr, rot, st = g.track(pose[:3, 3], pose[:3, :3], err_surface=err, **opts)
    np.testing.assert_allclose(tr, otr, atol=1e-6, rtol=0)
    np.testing.assert_allclose(rot, orot, atol=1e-6, rtol=0)
    assert st.so3_iterations == ost.so3_iterations
    if not opts.get("rgb_only"):
        assert st.last_icp_count == ost.last_icp_count
        np.testing.assert_allclose(st.last_icp_error, ost.last_icp_error, rtol=1e-6)
        _eq(err.cpu().numpy(), oerr, "ICP error surface after tracking")
    assert st.last_rgb_count == ost.last_rgb_count
    np.testing.assert_allclose(np.array(st.lastA), np.array(ost.lastA), rtol=1e-9, atol=1e-12)
    exact = np.array_equal(tr, otr) and np.array_equal(rot, orot)
    print("pose bit-exact:", exact)
    g.close()


def test_screen_box_culling_leaves_the_gauss_newton_loop_bit_identical():
    """cf_odom_set_culling: an object-sized model (the prediction cut down to a rectangle) tracked with the occupancy look-up and the
    screen-box culling of the ICP reduction -- workgroups outside the re-projected bounding box of the predicted vertices leave before
    they load anything -- must give the bits of the unculled run and of the oracle; the box must be a proper part of the image,
    contain the model's rectangle, and an empty prediction must cull everything (exact zeros, pose unchanged)."""
    from co_fusion_amd import api
    W, H = 320, 240
    fp = common.frame_pair(W, H, noise=True)
    cam = fp["cam"]
    pose = common.perturbed_pose(2)
    x0, x1, y0, y1 = 120, 200, 60, 150
    v4 = fp["v4"].copy(); n4 = fp["n4"].copy()
    keep = np.zeros((H, W), bool); keep[y0:y1, x0:x1] = True
    v4[~keep] = 0; n4[~keep] = 0
    ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
    d = ctx.to_device

    def run(vv, nn, cull):
        g = api.Odometry(ctx)
        g.set_culling(cull)
        g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(vv), d(nn), pose); g.init_rgb_model(d(fp["img"]))
        g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
        err = ctx.empty((H, W))
        tr, rot, st = g.track(pose[:3, 3], pose[:3, :3], err_surface=err)
        e = err.cpu().numpy()
        g.close()
        return tr, rot, st, e

    t0, r0, s0, e0 = run(v4, n4, False)
    t1, r1, s1, e1 = run(v4, n4, True)
    assert t0.tobytes() == t1.tobytes() and r0.tobytes() == r1.tobytes() and e0.tobytes() == e1.tobytes()
    assert s0.last_icp_count == s1.last_icp_count > 100 and s0.last_rgb_count == s1.last_rgb_count
    assert np.array_equal(np.array(s0.lastA), np.array(s1.lastA)) and np.array_equal(np.array(s0.lastb), np.array(s1.lastb))
    b = list(s1.cull_box)
    assert list(s0.cull_box) == [0, 0, W - 1, H - 1]
    assert b[0] <= x0 and b[1] <= y0 and b[2] >= x1 - 1 and b[3] >= y1 - 1, b
    assert (min(b[2], W - 1) - max(b[0], 0) + 1) * (min(b[3], H - 1) - max(b[1], 0) + 1) < 0.8 * W * H, f"the box {b} culls nothing"
    # the oracle on the same cut-down prediction
    od = orc.Odometry(W, H, cam.cx, cam.cy, cam.fx, cam.fy)
    od.init_first_rgb(fp["rgba0"]); od.init_icp_model(v4, n4, pose); od.init_rgb_model(fp["img"])
    od.init_icp(orc.depth_pyramid(fp["d1"]), 20.0); od.init_rgb(fp["rgba1"])
    otr, orot, ost = od.track(pose[:3, 3], pose[:3, :3])
    assert t1.tobytes() == np.asarray(otr, np.float32).tobytes() and r1.tobytes() == np.asarray(orot, np.float32).tobytes()
    assert s1.last_icp_count == ost.last_icp_count
    # a SECOND tracking call on the same preparation (retries, A/B loops): the bounding-box accumulator was latched and zeroed by the first
    # call, so the second one must fall back to the whole image instead of reading an empty box and culling every workgroup
    def run_twice(cull):
        g = api.Odometry(ctx)
        g.set_culling(cull)
        g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(v4), d(n4), pose); g.init_rgb_model(d(fp["img"]))
        g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
        g.track(pose[:3, 3], pose[:3, :3])
        out = g.track(pose[:3, 3], pose[:3, :3])
        g.close()
        return out
    ta, ra, sa = run_twice(False)
    tb, rb, sb = run_twice(True)
    assert ta.tobytes() == tb.tobytes() and ra.tobytes() == rb.tobytes() and sa.last_icp_count == sb.last_icp_count > 100
    assert list(sb.cull_box) == [0, 0, W - 1, H - 1], "second call on one preparation: whole image"
    # a tracking call that starts from ANOTHER pose than the one the model maps were prepared with (a motion-model guess): the
    # screen box is the projection of a frustum piece expressed in the preparation's camera, so the library must not use it -- same bits as
    # without culling, and the box reported is the whole image
    def run_from(start, cull):
        g = api.Odometry(ctx)
        g.set_culling(cull)
        g.init_first_rgb(d(fp["rgba0"])); g.init_icp_model(d(v4), d(n4), pose); g.init_rgb_model(d(fp["img"]))
        g.init_icp(ctx.depth_pyramid(d(fp["d1"])), 20.0); g.init_rgb(d(fp["rgba1"]))
        out = g.track(start[:3, 3], start[:3, :3])
        g.close()
        return out
    start = pose.copy(); start[:3, 3] += np.array([0.02, -0.015, 0.01], np.float32)
    tc, rc, sc = run_from(start, False)
    td, rd, sd = run_from(start, True)
    assert tc.tobytes() == td.tobytes() and rc.tobytes() == rd.tobytes() and sc.last_icp_count == sd.last_icp_count > 100
    assert list(sd.cull_box) == [0, 0, W - 1, H - 1], "a call from another pose than the preparation's: no screen box"
    # an empty prediction: everything culled
    z4 = np.zeros_like(v4)
    t2, r2, s2, _ = run(z4, z4, True)
    t3, r3, s3, _ = run(z4, z4, False)
    assert t2.tobytes() == t3.tobytes() and r2.tobytes() == r3.tobytes() and s2.last_icp_count == s3.last_icp_count == 0
    assert s2.cull_box[0] > s2.cull_box[2], list(s2.cull_box)
    ctx.close()
