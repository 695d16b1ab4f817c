"""Shared definition of the reference pin cases (inputs + which function produces what), used by
tests/golden/make_ref_golden.py (runs the reference's own kernels, oracle/_ref) and by the pin tests (oracle on CPU, HIP on GPU).

Everything is derived from ONE seeded synthetic frame pair at 160x120, treated as a full-resolution frame with a 3-level pyramid,
so that the whole fixture (inputs included, so that it does not depend on the generator's RNG) stays small."""
from __future__ import annotations

import numpy as np

W, H = 160, 120
FX, FY, CX, CY = 132.0, 132.0, 80.0, 60.0
DEPTH_CUTOFF = 20.0
DIST_THRES, ANGLE_THRES = 0.10, np.float32(np.sin(20.0 * 3.14159254 / 180.0))  # RGBDOdometry.cpp:38-39
SOBEL_SCALE, MAX_DEPTH_DELTA = 0.125, 0.07
MIN_GRAD = (5.0, 3.0, 1.0)
SIGMAS = ("count", 1.0, -1.0)  # what rgbStep can be handed: the correspondence count, 1 (zero residual), -1 (rgbOnly)


def inputs(w=W, h=H):
    """Seeded inputs.  Returns a dict of arrays (stored verbatim in the small fixture, as digests in the full-size one)."""
    import common
    fp = common.frame_pair(w, h, noise=True)
    clean = common.frame_pair(w, h, noise=False)
    # depth: the sensor model's drop-outs (zeros) + a tenth of its noise, so that most pixels stay ICP inliers at this resolution
    dn = fp["d1"].astype(np.float32); dc = clean["d1"].astype(np.float32)
    fp = dict(fp); fp["d1"] = np.where(dn > 0, dc + np.float32(0.1) * (dn - dc), 0).astype(np.float32)
    fp["v4"], fp["n4"], fp["img"] = clean["v4"], clean["n4"], clean["img"]  # model prediction: rendered from the noise-free scene
    pose = common.perturbed_pose(3)
    T2 = (common.perturbed_pose(7, 0.004, 0.3) @ pose).astype(np.float32)
    dT = common.perturbed_pose(11, 0.003, 0.2).astype(np.float32)
    Rr = common.perturbed_pose(5, 0.0, 0.4)[:3, :3].astype(np.float32)
    s = w / float(W)
    return dict(cam=np.array([FX * s, FY * s, CX * s, CY * s], np.float32), d1=fp["d1"].astype(np.float32), rgba0=fp["rgba0"], rgba1=fp["rgba1"], v4=fp["v4"].astype(np.float32),
                n4=fp["n4"].astype(np.float32), img=fp["img"], pose=pose.astype(np.float32), T2=T2, dT=dT, Rr=Rr)


def run(m, inp, cam_cls):
    """Run every pinned function of module `m` (tests/ref.py = reference kernels, tests/orc.py = oracle; same names) on the
    inputs.  Returns {name: array}.  Reductions are returned as the reference returns them (A, b, residual as f32)."""
    out = {}
    fx, fy, cx, cy = [float(v) for v in inp["cam"]] if "cam" in inp else (FX, FY, CX, CY)
    cam = cam_cls(fx, fy, cx, cy)
    is_ref = m.__name__.endswith("ref")

    def level(c, l):
        d = float(1 << l)
        return cam_cls(c.fx / d, c.fy / d, c.cx / d, c.cy / d)

    # ---- frame-side preparation (RGBDOdometry::initICP / initRGB, §8 rows a1-a3)
    dep = [inp["d1"]]
    for l in range(2):
        dep.append(m.pyrdown_gauss_f32(dep[l]))
    out["depth_pyr1"], out["depth_pyr2"] = dep[1], dep[2]
    vmaps, nmaps = [], []
    for l in range(3):
        v = m.create_vmap(dep[l], level(cam, l), DEPTH_CUTOFF)
        n = m.create_nmap(v)
        vmaps.append(v); nmaps.append(n)
        out[f"vmap{l}"], out[f"nmap{l}"] = v, n
    # ---- model-side preparation (initICPModel)
    pv, pn = m.copy_maps(inp["v4"], inp["n4"])
    out["copy_v"], out["copy_n"] = pv, pn
    mv, mn = [pv], [pn]
    for l in range(2):
        mv.append(m.resize_map(mv[l], False)); mn.append(m.resize_map(mn[l], True))
        out[f"resize_v{l + 1}"], out[f"resize_n{l + 1}"] = mv[l + 1], mn[l + 1]
    pose = inp["pose"]
    for l in range(3):
        mv[l], mn[l] = m.transform_maps(mv[l], mn[l], pose[:3, :3], pose[:3, 3])
        out[f"model_v{l}"], out[f"model_n{l}"] = mv[l], mn[l]
    # ---- RGB side (initRGBModel / initRGB / computeDerivativeImages / projectToPointCloud)
    model_depth = [m.vertices_to_depth(inp["v4"], DEPTH_CUTOFF)]
    last_img = [m.rgba_to_intensity(inp["img"])]
    next_img = [m.rgba_to_intensity(inp["rgba1"])]
    first_img = [m.rgba_to_intensity(inp["rgba0"])]
    for l in range(2):
        model_depth.append(m.pyrdown_gauss_f32(model_depth[l]))
        last_img.append(m.pyrdown_gauss_u8(last_img[l])); next_img.append(m.pyrdown_gauss_u8(next_img[l]))
        first_img.append(m.pyrdown_gauss_u8(first_img[l]))
    out["model_depth0"], out["model_depth2"] = model_depth[0], model_depth[2]
    out["next_img0"], out["next_img2"] = next_img[0], next_img[2]
    # ---- reductions, every level
    Rprev = pose[:3, :3]; tprev = pose[:3, 3]
    Rprev_inv = np.linalg.inv(Rprev.astype(np.float64)).astype(np.float32)
    T2 = inp["T2"]
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
    dT = inp["dT"].astype(np.float64)
    for l in range(3):
        cl = level(cam, l)
        if is_ref:
            A, b, res, err = m.icp_step(T2[:3, :3], T2[:3, 3], vmaps[l], nmaps[l], Rprev_inv, tprev, cl, mv[l], mn[l], DIST_THRES,
                                        ANGLE_THRES, want_err=True)
        else:
            sums, err = m.icp_step(T2[:3, :3], T2[:3, 3], vmaps[l], nmaps[l], Rprev_inv, tprev, cl, mv[l], mn[l], DIST_THRES,
                                   ANGLE_THRES, want_err=True)
            A, b, res = m.icp_sums_to_host(sums) if hasattr(m, "icp_sums_to_host") else m.se3_to_host(sums)  # (follows orc_set_icp_arith)
            tA, tb, tres = m.icp_step_ref_order(T2[:3, :3], T2[:3, 3], vmaps[l], nmaps[l], Rprev_inv, tprev, cl, mv[l], mn[l],
                                                DIST_THRES, ANGLE_THRES)
            out[f"icp_A{l}_order"], out[f"icp_b{l}_order"], out[f"icp_res{l}_order"] = tA, tb, tres
        out[f"icp_A{l}"], out[f"icp_b{l}"], out[f"icp_res{l}"], out[f"icp_err{l}"] = A, b, res, err
        dx, dy = m.sobel(next_img[l])
        out[f"dIdx{l}"], out[f"dIdy{l}"] = dx, dy
        Kl = K.copy(); Kl[:2] /= (1 << l)
        krkinv = (Kl @ dT[:3, :3] @ np.linalg.inv(Kl)).astype(np.float32)
        kt = (Kl @ dT[:3, 3]).astype(np.float32)
        min_scale = float(MIN_GRAD[l] ** 2 / SOBEL_SCALE ** 2)
        r = m.rgb_residual(min_scale, dx, dy, model_depth[l], model_depth[l], last_img[l], next_img[l], MAX_DEPTH_DELTA, kt, krkinv)
        cor, sig, cnt = r[0], r[1], r[2]
        valid = cor["valid"] != 0
        out[f"corres_valid{l}"] = valid.astype(np.uint8)
        for f in ("zero_x", "zero_y", "one_x", "one_y", "diff"):
            out[f"corres_{f}{l}"] = np.where(valid, cor[f], 0).astype(cor[f].dtype)
        out[f"residual_sigma_count{l}"] = np.array([sig, cnt], np.int64)
        cloud = m.project_cloud(model_depth[l], cam, l) if is_ref else m.project_cloud(model_depth[l], cl)
        out[f"cloud{l}"] = cloud
        for s in SIGMAS:
            sigma = float(cnt) if s == "count" else float(s)
            if is_ref:
                A, b = m.rgb_step(cor, sigma, cloud, cl.fx, cl.fy, dx, dy, SOBEL_SCALE)
            else:
                sums = m.rgb_step(cor, sigma, cloud, cl.fx, cl.fy, dx, dy, SOBEL_SCALE)
                A, b, _ = m.se3_to_host(sums, m.rgb_fix_bits(sigma))
                tA, tb = m.rgb_step_ref_order(cor, sigma, cloud, cl.fx, cl.fy, dx, dy, SOBEL_SCALE)
                out[f"rgb_A{l}_{s}_order"], out[f"rgb_b{l}_{s}_order"] = tA, tb
            out[f"rgb_A{l}_{s}"], out[f"rgb_b{l}_{s}"] = A, b
    # ---- SO3 pre-alignment on the coarsest level
    K2 = K.copy(); K2[:2] /= 4
    Rr = inp["Rr"].astype(np.float64)
    basis = (K2 @ Rr @ np.linalg.inv(K2)).astype(np.float32)
    kinv = np.linalg.inv(K2).astype(np.float32)
    krlr = (K2 @ Rr).astype(np.float32)
    if is_ref:
        A, b, res = m.so3_step(first_img[2], next_img[2], basis, kinv, krlr)
    else:
        A, b, res = m.so3_to_host(m.so3_step(first_img[2], next_img[2], basis, kinv, krlr))
        out["so3_A_order"], out["so3_b_order"], out["so3_res_order"] = m.so3_step_ref_order(first_img[2], next_img[2], basis, kinv, krlr)
    out["so3_A"], out["so3_b"], out["so3_res"] = A, b, res
    return out


REDUCTION_PREFIXES = ("icp_A", "icp_b", "rgb_A", "rgb_b", "so3_A", "so3_b")


def is_reduction(name: str) -> bool:
    return name.startswith(REDUCTION_PREFIXES)


def bits_equal(a, b) -> bool:
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind == "f":
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


# f32 sums in tree order vs exact sums: a few ulp of the LARGEST entry (entries with cancellation are compared on that scale)
SUM_RTOL = 2e-5


def sums_close(a, b) -> bool:
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return bool(np.abs(a - b).max() <= SUM_RTOL * max(np.abs(b).max(), 1e-30))


# =====================================================================================================================
# surfel passes: one scripted three-frame scenario, run by three back ends
#   * the reference's own GLSL shaders (tests/ref.py surfel_passes -> oracle/_ref)        -> tests/golden/ref_surfel_v1.npz
#   * the CPU oracle (tests/orc_pipeline.py)                                             -> tests/test_cpu_refpin.py
#   * the HIP path through the C-ABI (co_fusion_amd.model.Model)                         -> tests/test_refpin_gpu.py
# =====================================================================================================================
import hashlib

SW, SH = 160, 120
MAX_DEPTH, DEPTH_FILTER_CUTOFF, TIME_DELTA = 20.0, 5.0, 200
CONF_SPLAT, CONF_CLEAN, OUTLIER_COEFF = 0.0, 0.9, 3.0


def digest(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = np.where(np.isnan(a), np.float32(np.nan), a).astype(a.dtype)  # one NaN encoding
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8).copy()


def surfel_inputs(w=SW, h=SH):
    import common
    from co_fusion_amd import synth
    cam = synth.Camera.scaled(w, h)
    sc = synth.Scene(n_obj=1)
    out = dict(cam=np.array([cam.fx, cam.fy, cam.cx, cam.cy], np.float32))
    for k, t in enumerate((0, 2, 4)):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        d = d.astype(np.float32)
        if k == 1:
            d[h // 12:h // 6, w * 3 // 16:w * 3 // 8] = 0.0  # holes
            d[h * 5 // 6, w * 5 // 8] = 7.0                  # beyond the filter cutoff
        out[f"d{k}"] = d; out[f"rgba{k}"] = synth.rgb_to_rgba(rgb)
    mask = np.zeros((h, w), np.uint8); mask[:, w // 2:] = 1
    out["mask"] = mask
    out["pose1"] = common.perturbed_pose(4, 0.003, 0.2).astype(np.float32)
    out["pose2"] = (common.perturbed_pose(6, 0.004, 0.3) @ common.perturbed_pose(4, 0.003, 0.2)).astype(np.float32)
    return out


class CpuSurfelBackend:
    """tests/orc_pipeline.py functions (oracle, or the reference's shaders inside ref.surfel_passes())."""

    def __init__(self, op, cam):
        import orc
        self.op = op; self.cam = orc.Cam(*[float(v) for v in cam]); self.surfels = None; self.w = self.h = None

    def bilateral(self, d): return self.op.bilateral(d, DEPTH_FILTER_CUTOFF)

    def bootstrap(self, rgba, d, df):
        self.h, self.w = d.shape
        raw, n = self.op.vertex_feedback(rgba, d, self.cam, 1, MAX_DEPTH)
        filt, _ = self.op.vertex_feedback(rgba, df, self.cam, 1, MAX_DEPTH)
        self.surfels = self.op.model_initialise(raw, n, filt)

    def map(self): return self.surfels

    def predict_indices(self, pose, time):
        self.idx = self.op.predict_indices(self.surfels, pose, self.cam, self.w, self.h, MAX_DEPTH, time, TIME_DELTA)
        return self.idx

    def combined_predict(self, pose, time):
        self.pred = self.op.combined_predict(self.surfels, pose, self.cam, self.w, self.h, MAX_DEPTH, CONF_SPLAT, time, time, TIME_DELTA)
        return self.pred

    def fill_in(self, rgba, df, pg, pr):
        img, vc, nr, _ = self.pred
        return self.op.fill_in(vc, nr, img, df, rgba, self.cam, pg, pr)

    def fuse(self, pose, time, rgba, mask, d, df, weighting, mask_id):
        idx, vc, ct, nr = self.idx
        self.surfels, self.new = self.op.fuse(self.surfels, idx, vc, nr, rgba, d, df, mask, pose, self.cam, time, weighting, mask_id, MAX_DEPTH)
        return np.concatenate([self.surfels, self.new])

    def clean(self, pose, time, df, mask, mask_id):
        idx, vc, ct, nr = self.idx
        self.surfels = self.op.clean(self.surfels, self.new, idx, vc, ct, df, mask, pose, self.cam, time, CONF_CLEAN, OUTLIER_COEFF,
                                     TIME_DELTA, mask_id)
        return self.surfels


def surfel_run(be, inp):
    """the scenario; returns {name: array} (every array is a pinned output)"""
    out = {}
    df = [be.bilateral(inp[f"d{k}"]) for k in range(3)]
    for k in range(3):
        out[f"bilateral{k}"] = df[k]
    be.bootstrap(inp["rgba0"], inp["d0"], df[0])
    out["bootstrap_map"] = be.map()
    for tick, k, pose, mask_id, w in ((2, 1, inp["pose1"], 0, 0.8), (3, 2, inp["pose2"], 1, 1.0)):
        rgba, d = inp[f"rgba{k}"], inp[f"d{k}"]
        for name, a in zip(("image", "vertexConf", "normalRad", "time"), be.combined_predict(pose, tick)):
            out[f"t{tick}_splat_{name}"] = a
        for pg, pr in ((False, False), (True, True)):
            for name, a in zip(("vertex", "normal", "image"), be.fill_in(rgba, df[k], pg, pr)):
                out[f"t{tick}_fill{int(pg)}_{name}"] = a
        for name, a in zip(("index", "vertConf", "colorTime", "normRad"), be.predict_indices(pose, tick)):
            out[f"t{tick}_index_{name}"] = a
        out[f"t{tick}_fuse_map"] = be.fuse(pose, tick, rgba, inp["mask"], d, df[k], w, mask_id)
        be.predict_indices(pose, tick)
        out[f"t{tick}_clean_map"] = be.clean(pose, tick, df[k], inp["mask"], mask_id)
    return out


def surfel_summary(out):
    """small integers that make a degenerate scenario visible in the fixture"""
    return np.array([out["bootstrap_map"].shape[0], int((out["t2_index_index"] > 0).sum()), int((out["t2_splat_vertexConf"][..., 2] > 0).sum()),
                     out["t2_fuse_map"].shape[0], int((out["t2_fuse_map"][:, 7] == 2).sum()), out["t2_clean_map"].shape[0],
                     out["t3_fuse_map"].shape[0], int((out["t3_fuse_map"][:, 7] == 3).sum()), out["t3_clean_map"].shape[0]], np.int64)
