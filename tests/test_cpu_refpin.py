"""Pins the CPU oracle (oracle/*.c) against the REFERENCE'S OWN tracking kernels.

tests/golden/ref_v1.npz holds the outputs of Core/Cuda/reduce.cu + cudafuncs.cu, compiled from /root/reference with g++ under the
CPU SIMT emulator of oracle/ref_shim and executed on seeded inputs (tests/golden/make_ref_golden.py).  Bars:
  * per-pixel outputs (vertex / normal maps, pyramids, intensity, Sobel, clouds, DataTerm records, ICP error surface) and integer
    results (correspondence count, sigma sum, inlier counts): bit-exact;
  * normal-equation sums: the oracle evaluated in the reference's f32 summation order (thread-strided partials, shuffle tree,
    second-stage reduceSum) is bit-exact; the order-independent exact fixed-point sums the HIP path uses agree to f32 rounding
    of that tree (refpin.SUM_RTOL of the largest entry)."""
import os
import subprocess

import numpy as np
import pytest

import orc
import ref
import refpin

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_v1.npz")


@pytest.fixture(scope="module")
def golden():
    z = np.load(GOLDEN)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("ref_")}
    return inp, out


def test_oracle_matches_reference_kernels(golden):
    inp, want = golden
    got = refpin.run(orc, inp, orc.Cam)
    assert want["residual_sigma_count0"][1] > 500 and want["icp_res0"][1] > 0.5 * refpin.W * refpin.H, "degenerate pin scene"
    checked = 0
    for name, w in want.items():
        g = got[name]
        if refpin.is_reduction(name):
            assert refpin.bits_equal(got[name + "_order"], w), f"{name}: oracle in reference summation order differs"
            assert refpin.sums_close(g, w), f"{name}: exact sums vs reference f32 tree: {np.abs(g - w).max()}"
        elif name.startswith(("icp_res", "so3_res")):
            assert refpin.bits_equal(got[name + "_order"], w), name
            assert g[1] == w[1] and refpin.sums_close(g[:1], w[:1]), name  # [sum r^2, inlier count]
        else:
            assert refpin.bits_equal(g, w), f"{name}: differs from the reference kernel's output"
        checked += 1
    assert checked == len(want) >= 80


def test_gram_form_of_the_icp_sums_is_pinned_to_the_reference_too(golden):
    """ORC_ICP_ARITH_GRAM (row entries rounded once, exact integer Gram matrix; the HIP kernels' matrix-core option, cf_set_icp_arith):
    against the reference kernels' f32 tree to the same bar as the product form, identical inlier counts and error surfaces, and within
    1e-6 of the largest entry of the product form."""
    inp, want = golden
    prod = refpin.run(orc, inp, orc.Cam)
    orc.set_icp_arith("gram")
    try:
        got = refpin.run(orc, inp, orc.Cam)
    finally:
        orc.set_icp_arith("product")
    for l in range(3):
        for name in (f"icp_A{l}", f"icp_b{l}"):
            assert refpin.sums_close(got[name], want[name]), f"{name}: Gram sums vs reference f32 tree: {np.abs(got[name] - want[name]).max()}"
            assert not refpin.bits_equal(got[name], prod[name]) or l == 2, f"{name}: the Gram form should differ from the product form in the last bits"
            assert np.abs(got[name] - prod[name]).max() <= 1e-6 * max(np.abs(prod[f"icp_A{l}"]).max(), 1.0), name
        g, w = got[f"icp_res{l}"], want[f"icp_res{l}"]
        assert g[1] == w[1] and refpin.sums_close(g[:1], w[:1]), f"icp_res{l}"
        assert refpin.bits_equal(got[f"icp_err{l}"], want[f"icp_err{l}"])
    for name in want:   # everything that is not an ICP sum is untouched by the switch
        if not name.startswith(("icp_A", "icp_b", "icp_res")):
            assert refpin.bits_equal(got[name], prod[name]), name


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_fixture_is_what_the_reference_library_produces(golden):
    """The committed fixture is reproducible from the reference sources (spot check: preparation + one reduction)."""
    inp, want = golden
    cam = orc.Cam(refpin.FX, refpin.FY, refpin.CX, refpin.CY)
    v = ref.create_vmap(inp["d1"], cam, refpin.DEPTH_CUTOFF)
    assert refpin.bits_equal(v, want["vmap0"])
    assert refpin.bits_equal(ref.create_nmap(v), want["nmap0"])
    img = ref.rgba_to_intensity(inp["rgba1"])
    assert refpin.bits_equal(img, want["next_img0"])
    dx, dy = ref.sobel(img)
    assert refpin.bits_equal(dx, want["dIdx0"]) and refpin.bits_equal(dy, want["dIdy0"])
    pose = inp["pose"]; T2 = inp["T2"]
    A, b, res, err = ref.icp_step(T2[:3, :3], T2[:3, 3], want["vmap2"], want["nmap2"],
                                  np.linalg.inv(pose[:3, :3].astype(np.float64)).astype(np.float32), pose[:3, 3],
                                  cam.level(2), want["model_v2"], want["model_n2"], refpin.DIST_THRES, refpin.ANGLE_THRES, want_err=True)
    assert refpin.bits_equal(A, want["icp_A2"]) and refpin.bits_equal(b, want["icp_b2"]) and refpin.bits_equal(res, want["icp_res2"])
    assert refpin.bits_equal(err, want["icp_err2"])


def test_rgb_fixed_point_window_follows_sigma():
    """rgbOnly (sigma = -1) and zero-residual (sigma = 1) rows are ~count times larger than regular ones: the fixed-point window
    must move with sigma (a fixed Q32 wrapped there; found by the reference pin)."""
    assert orc.rgb_fix_bits(-1.0) == 8 and orc.rgb_fix_bits(1.0) == 8 and orc.rgb_fix_bits(0.0) == 8
    assert orc.rgb_fix_bits(2.0) == 10 and orc.rgb_fix_bits(255.0) == 22 and orc.rgb_fix_bits(256.0) == 24
    assert orc.rgb_fix_bits(4095.0) == 30 and orc.rgb_fix_bits(4096.0) == 32 and orc.rgb_fix_bits(3.0e5) == 32


# ---- surfel passes: oracle vs the reference's own GLSL shaders ---------------------------------------------------------------
SURFEL_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_surfel_v1.npz")


def _surfel_golden():
    z = np.load(SURFEL_GOLDEN)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    sha = {k[4:]: z[k] for k in z.files if k.startswith("sha_")}
    shape = {k[6:]: tuple(z[k]) for k in z.files if k.startswith("shape_")}
    return inp, sha, shape, z["summary"]


def test_oracle_surfel_passes_match_reference_shaders():
    """bilateral, bootstrap, index map, splat prediction, fill-in, fuse, clean over three frames: every output array of the oracle
    has the sha256 of what the reference's shader sources produced (bit-exact)."""
    import orc_pipeline as op
    inp, sha, shape, summary = _surfel_golden()
    assert summary[4] > 1000 and summary[7] > 1000, "the merge path must be exercised"
    assert summary[2] > 0.5 * refpin.SW * refpin.SH and summary[1] > 0.5 * refpin.SW * refpin.SH
    got = refpin.surfel_run(refpin.CpuSurfelBackend(op, inp["cam"]), inp)
    assert set(got) == set(sha) and len(sha) >= 36
    assert np.array_equal(refpin.surfel_summary(got), summary)
    for name, a in got.items():
        assert tuple(np.asarray(a).shape) == shape[name], f"{name}: shape {np.asarray(a).shape} vs reference {shape[name]}"
        assert np.array_equal(refpin.digest(a), sha[name]), f"{name}: differs from the reference shaders' output"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_surfel_fixture_is_what_the_reference_shaders_produce():
    import orc_pipeline as op
    inp, sha, shape, summary = _surfel_golden()
    with ref.surfel_passes() as rop:
        want = refpin.surfel_run(refpin.CpuSurfelBackend(rop, inp["cam"]), inp)
    got = refpin.surfel_run(refpin.CpuSurfelBackend(op, inp["cam"]), inp)
    for name, w in want.items():
        assert np.array_equal(refpin.digest(w), sha[name]), f"{name}: fixture is stale"
        g = np.asarray(got[name]); w = np.asarray(w)
        assert g.shape == w.shape, name
        same = (g == w) | ((g != g) & (w != w)) if g.dtype.kind == "f" else (g == w)
        assert same.all(), f"{name}: {np.count_nonzero(~same)} of {same.size} values differ from the reference shaders"


# ---- connected components of the segmentation post-processing: the reference's header compiled as is ------------------------
@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_connected_labels_match_reference_header():
    """Core/Segmentation/ConnectedLabels.hpp (header-only, compiled against a cv::Mat stand-in) vs oracle/orc_segment.c:
    component image and {label, bbox, size} per component identical, on label images from blobby to salt-and-pepper."""
    import ctypes as C
    rng = np.random.default_rng(7)
    cases = []
    for (h, w, nlab, smooth) in ((30, 40, 3, 4), (30, 40, 5, 1), (48, 64, 2, 8), (7, 5, 4, 1), (1, 9, 3, 1), (9, 1, 2, 1), (30, 40, 1, 1)):
        a = rng.integers(0, nlab, size=((h + smooth - 1) // smooth, (w + smooth - 1) // smooth), dtype=np.uint8)
        a = np.kron(a, np.ones((smooth, smooth), np.uint8))[:h, :w]
        a[rng.random(a.shape) < 0.05] = 255  # the "rejected" label
        cases.append(np.ascontiguousarray(a))
    for a in cases:
        h, w = a.shape
        res = []
        for fn in (orc.lib.orc_connected_labels, ref.lib().ref_connected_labels):
            comp = np.zeros((h, w), np.int32); st = np.zeros((h * w, 6), np.int32)
            n = fn(a.ctypes.data_as(C.c_void_p), w, h, comp.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), h * w)
            res.append((n, comp, st[:n].copy()))
        assert res[0][0] == res[1][0], f"{h}x{w}: component count"
        assert np.array_equal(res[0][1], res[1][1]), f"{h}x{w}: component image"
        assert np.array_equal(res[0][2], res[1][2]), f"{h}x{w}: component statistics"


# ---- both pins again at BASELINE.json's frame size --------------------------------------------------------------------------
FULL_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_full_v1.npz")
PLANAR = ("vmap", "nmap", "copy_", "resize_", "model_v", "model_n")


def _planar_valid_only(m):
    m = m.copy()
    h = m.shape[0] // 3
    bad = np.isnan(m[:h])
    m[h:2 * h][bad] = 0; m[2 * h:][bad] = 0
    return m


def test_oracle_matches_reference_at_full_resolution():
    """640x480: inputs regenerated from the seeded generator (checked against stored digests), outputs of the oracle against the
    digests / values the reference's kernels and shaders produced (tests/golden/make_ref_full_golden.py)."""
    import orc_pipeline as op
    z = np.load(FULL_GOLDEN)
    w, h = int(z["size"][0]), int(z["size"][1])
    assert (w, h) == (640, 480)
    inp = refpin.inputs(w, h)
    for k, v in inp.items():
        assert np.array_equal(refpin.digest(v), z["insha_" + k]), f"input generator drifted: {k}"
    got = refpin.run(orc, inp, orc.Cam)
    n = 0
    for name, g in got.items():
        g = np.asarray(g)
        if name.endswith("_order"):
            continue
        if "val_" + name in z.files:
            wv = z["val_" + name]
            if refpin.is_reduction(name) or name.startswith(("icp_res", "so3_res")):
                assert refpin.bits_equal(np.asarray(got[name + "_order"]), wv), f"{name}: oracle in reference summation order differs"
                assert refpin.sums_close(g[:1] if "res" in name else g, wv[:1] if "res" in name else wv), name
            else:
                assert np.array_equal(g.astype(wv.dtype), wv), name
        else:
            gg = _planar_valid_only(g) if name.startswith(PLANAR) else g
            assert np.array_equal(refpin.digest(gg), z["sha_" + name]), f"{name}: differs from the reference kernel's output at {w}x{h}"
        n += 1
    assert n >= 80
    sinp = refpin.surfel_inputs(w, h)
    for k, v in sinp.items():
        assert np.array_equal(refpin.digest(v), z["sinsha_" + k]), f"input generator drifted: {k}"
    sgot = refpin.surfel_run(refpin.CpuSurfelBackend(op, sinp["cam"]), sinp)
    assert np.array_equal(refpin.surfel_summary(sgot), z["ssummary"])
    for name, a in sgot.items():
        assert np.array_equal(refpin.digest(a), z["ssha_" + name]), f"{name}: differs from the reference shaders' output at {w}x{h}"


# ---- the segmentation stage against the reference's own Core/Segmentation sources -----------------------------------------------
SEG_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_seg_v1.npz")


def _seg_golden():
    z = np.load(SEG_GOLDEN)
    n = int(z["n_calls"][0])
    return [{k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(f"c{i}/")} for i in range(n)]


@pytest.fixture(scope="module")
def seg_calls():
    import segpin
    return segpin.capture()


def test_segmentation_matches_reference_sources(seg_calls):
    """Every performSegmentationCRF call of a seeded multi-object run (model spawning included): the oracle's label mask, super-pixel
    counts, bounding boxes and new-label decision equal those of the reference's own Segmentation.cpp / Slic.h / ConnectedLabels.hpp
    (tests/golden/ref_seg_v1.npz; gSLICr and densecrf, absent from the reference tree, are stood in for by the oracle's SLIC and exact
    mean-field operations); float statistics agree to the f32 rounding of the reference's running sums."""
    import segpin
    want = _seg_golden()
    assert len(seg_calls) == len(want) >= 8
    spawned = False
    for i, (c, w) in enumerate(zip(seg_calls, want)):
        assert segpin.input_digest(c) == str(w["input_sha"]), f"call {i}: the regenerated inputs are not the fixture's"
        segpin.compare(c["oracle"], w, f"call {i}")
        spawned = spawned or c["oracle"]["hasNewLabel"]
    assert spawned and max(len(c["ids"]) for c in seg_calls) >= 2, "the scenario must spawn an object model"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_segmentation_fixture_is_what_the_reference_sources_produce(seg_calls):
    import segpin
    want = _seg_golden()
    for i in (0, 3, len(seg_calls) - 1):
        got = segpin.pack_result(segpin.run_reference(seg_calls[i]))
        assert got["full_sha"] == str(want[i]["full_sha"]) and np.array_equal(got["ints"], want[i]["ints"])
        assert refpin.bits_equal(got["floats"], want[i]["floats"]) and refpin.bits_equal(got["depth_range"], want[i]["depth_range"])


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_ground_truth_mask_branch_matches_reference_sources():
    """Segmentation::performSegmentation with FrameData.mask (Segmentation.cpp:59-119): label remapping through the function-static
    table, one new label per call, per-model pixel counts / 256 and depth statistics -- reference sources vs oracle, call by call."""
    import ctypes as C
    import orc_multi as om
    import warnings
    from co_fusion_amd import synth
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    w, h = 160, 128
    cam = synth.Camera.scaled(w, h)
    sc = synth.Scene(n_obj=3)
    mapping = np.zeros(256, np.uint8)     # the oracle's copy of the reference's function-static table
    ids, next_id = [0], 1
    L = ref.lib()
    for t in range(5):
        d, _, lab, _ = sc.render(cam, t, noise=True)
        gt = (lab * 40).astype(np.uint8)
        allow = t % 2 == 0
        o = om.segment_gt(gt, d, ids, next_id, allow, mapping)
        n = len(ids)
        cids = (C.c_uint * n)(*ids)
        full = np.zeros((h, w), np.uint8); models = (om.SegModel * (n + 1))(); n_out = C.c_int(); has_new = C.c_int()
        L.ref_segment_gt(orc.P(gt), orc.P(orc.f32(d)), w, h, n, cids, C.c_uint(next_id), int(allow), orc.P(full), models, C.byref(n_out),
                         C.byref(has_new))
        assert np.array_equal(full, o["full"]), f"call {t}: remapped mask"
        assert bool(has_new.value) == o["hasNewLabel"] and n_out.value == len(o["modelData"]), f"call {t}: new label / model rows"
        for m, r in zip(models[:n_out.value], o["modelData"]):
            assert (m.id, m.superPixelCount) == (r["id"], r["superPixelCount"]), f"call {t}: id / count"
            assert refpin.bits_equal(np.float32([m.avgConfidence, m.depthMean, m.depthStd]),
                                     np.float32([r["avgConfidence"], r["depthMean"], r["depthStd"]])), f"call {t}: statistics of model {m.id}"
        if o["hasNewLabel"]:
            ids.append(next_id); next_id += 1
    assert len(ids) >= 3


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_requires_fill_in_matches_resize_shader():
    """CoFusion::requiresFillIn (CoFusion.cpp:547-565): resize.frag into the 20x smaller buffer + the count, reference shader vs oracle,
    on predictions with holes of every size (incl. the decision flipping around the 0.75 ratio)."""
    import ctypes as C
    rng = np.random.default_rng(11)
    L = ref.lib()
    flips = set()
    for (w, h) in ((640, 480), (320, 240), (160, 120)):
        for fill in (0.2, 0.6, 0.74, 0.76, 0.9, 1.0):
            img = rng.integers(1, 256, size=(h, w, 4), dtype=np.uint8)
            coarse = rng.random((h // 10, w // 10)) > fill
            hole = np.kron(coarse, np.ones((10, 10), bool))
            img[hole] = 0
            img[rng.random((h, w)) < 0.02, 1] = 0   # single zero channels count as holes too
            img = np.ascontiguousarray(img)
            a = orc.lib.orc_requires_fill_in(orc.P(img), w, h, C.c_float(0.75))
            b = L.ref_requires_fill_in(orc.P(img), w, h, C.c_float(0.75))
            assert a == b, f"{w}x{h} fill {fill}: oracle {a} vs reference shader {b}"
            flips.add(int(a))
    assert flips == {0, 1}


# ---- the Gauss-Newton loop: oracle vs the reference's own RGBDOdometry class -------------------------------------------------------
ODO_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_odo_v1.npz")
# getIncrementalTransformation runs 57 f32 tree reductions whose order the oracle does not follow (it sums exactly), and the stand-in
# Eigen states its own rounding conventions (oracle/ref_shim/eigen_fixed/Eigen/Core): poses agree to a few f32 ulps of the motion, not bits
ODO_POSE_TOL = 5e-6


ODO_FULL_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_odo_full_v1.npz")


def _odo_inputs(path=None):
    import hashlib
    import refodo
    z = np.load(path or ODO_GOLDEN)
    W, H, n_frames = (int(v) for v in z["meta"])
    cam, frames = refodo.record_tracking_inputs(W, H, n_frames)
    for fi in z["frames"]:
        fr = frames[int(fi)]
        h = hashlib.sha256()
        for k in ("prev_rgba", "v4", "n4", "pose", "img", "rgba"):
            h.update(np.ascontiguousarray(fr[k]).tobytes())
        for d in fr["depth_pyr"]:
            h.update(np.ascontiguousarray(d).tobytes())
        assert h.hexdigest() == str(z[f"f{int(fi)}/digest"]), "the recorded tracking inputs changed: regenerate tests/golden/ref_odo_v1.npz"
    return z, cam, frames, W, H


@pytest.mark.parametrize("arith", ["product", "gram"])
def test_gn_loop_matches_reference_odometry_class(arith):
    """(both rounding specifications of the ICP sums, orc_set_icp_arith)  SURVEY 8 row a7: orc_odom_get_incremental_transformation against RGBDOdometry::getIncrementalTransformation compiled from
    /root/reference (schedule, SO(3) pre-alignment loop, ICP/RGB weighting, LDL^T solve, computeUpdateSE3, pose composition,
    divergence guard), six option sets x two frames: identical inlier / correspondence counts, poses within ODO_POSE_TOL."""
    import refodo
    z, cam, frames, W, H = _odo_inputs()
    moved = 0.0
    orc.set_icp_arith(arith)
    try:
        _gn_loop_against_fixture(refodo, z, cam, frames, W, H, arith)
    finally:
        orc.set_icp_arith("product")


@pytest.mark.skipif(not os.path.exists(ODO_FULL_GOLDEN), reason="tests/golden/ref_odo_full_v1.npz not generated (make_ref_odo_golden.py full, ~1.5 h)")
def test_gn_loop_matches_reference_odometry_class_at_640x480():
    """the same pin at BASELINE.json's own frame size: RGBDOdometry::getIncrementalTransformation of the reference (its CUDA kernels under
    the emulator, ~20 minutes per call) on two recorded 640x480 frames x {default, fast_odom} against the oracle: identical counts, poses
    within ODO_POSE_TOL"""
    import refodo
    z, cam, frames, W, H = _odo_inputs(ODO_FULL_GOLDEN)
    assert (W, H) == (640, 480)
    _gn_loop_against_fixture(refodo, z, cam, frames, W, H, "product", options=[str(o) for o in z["options"]])


def _gn_loop_against_fixture(refodo, z, cam, frames, W, H, arith, options=None):
    moved = 0.0
    # f1/icp_only is a run that DIVERGES in the reference too (ICP alone slides 0.4 m along the wall on a 1 cm step): a chaotic
    # iteration, in which the product form happens to stay within ODO_POSE_TOL of the f32 tree and the Gram form's coarser row grid
    # (2^-20 / 2^-17 / 2^-22, bounded by the 64-bit accumulators) ends 1.5e-4 m away; the other eleven cases agree to 1e-7 in both forms
    loose = {"f1/icp_only": 5e-4} if arith == "gram" else {}
    for fi in z["frames"]:
        fr = frames[int(fi)]
        for opts in refodo.OPTION_SETS:
            if options is not None and opts[0] not in options:
                continue
            key = f"f{int(fi)}/{opts[0]}"
            tr, rot, st, err = refodo.track_once(orc.Odometry, cam, W, H, fr, opts)
            rt, rr, rs = z[key + "/trans"], z[key + "/rot"], z[key + "/stats"]
            tol = loose.get(key, ODO_POSE_TOL)
            assert np.abs(tr - rt).max() <= tol, f"{key}: translation {tr} vs reference {rt}"
            assert np.abs(rot - rr).max() <= tol, f"{key}: rotation differs by {np.abs(rot - rr).max()}"
            if key in loose:
                continue   # (statistics of a diverged run follow the pose)
            icp, rgb, so3 = not opts[1] and opts[2] > 0, opts[1] or opts[2] < 100, opts[5]
            if icp:
                assert st["last_icp_count"] == rs[1], f"{key}: ICP inliers {st['last_icp_count']} vs {rs[1]}"
                assert abs(st["last_icp_error"] - rs[0]) <= 1e-4 * rs[0], f"{key}: ICP error"
            if rgb:
                assert st["last_rgb_count"] == rs[3], f"{key}: RGB correspondences {st['last_rgb_count']} vs {rs[3]}"
                assert abs(st["last_rgb_error"] - rs[2]) <= 1e-4 * max(rs[2], 1e-6), f"{key}: RGB error"
            if so3:  # (with so3 off the reference's members keep their constructor values, the oracle reports zeros)
                assert st["last_so3_count"] == rs[5] and abs(st["last_so3_error"] - rs[4]) <= 1e-4 * rs[4], f"{key}: SO3 statistics"
            A, b = z[key + "/lastA"], z[key + "/lastb"]
            assert np.abs(st["lastA"] - A).max() <= 2e-4 * np.abs(A).max(), f"{key}: last normal matrix"
            assert np.abs(st["lastb"] - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-3 * np.abs(A).max()), f"{key}: last right-hand side"
            if icp:
                es = z[key + "/err_sum_max"]
                assert abs(err.astype(np.float64).sum() - es[0]) <= 1e-4 * es[0] and abs(err.max() - es[1]) <= 1e-4, f"{key}: ICP error surface"
            moved = max(moved, float(np.abs(tr - fr["pose"][:3, 3]).max()))
    assert moved > 5e-3, "degenerate pin: the tracker did not move"


@pytest.mark.parametrize("fixture", ["small", "640x480"])
def test_reference_order_gn_loop_is_the_reference_class_bit_for_bit(fixture):
    """under ORC_ICP_ARITH_REFERENCE the oracle's Gauss-Newton loop runs on the f32 trees of reduce.cu:90-185 at
    GPUConfig.h's launch shapes (orc_*_step_f32tree, each pinned to the reference KERNEL bit for bit above) with the host algebra in the
    order of the classes the reference's text instantiates -- and returns what RGBDOdometry::getIncrementalTransformation compiled from
    /root/reference returned for the same recorded inputs BIT FOR BIT: translation, rotation, the last normal equations in f64, every
    statistic; six option sets x two frames at 160x120, {default, fast_odom} x two frames at 640x480.  No tolerance (the product / Gram
    forms above: ODO_POSE_TOL)."""
    import refodo
    if fixture == "640x480" and not os.path.exists(ODO_FULL_GOLDEN):
        pytest.skip("tests/golden/ref_odo_full_v1.npz not generated")
    z, cam, frames, W, H = _odo_inputs(ODO_FULL_GOLDEN) if fixture == "640x480" else _odo_inputs()
    orc.set_icp_arith("reference")
    checked = 0
    try:
        for fi in z["frames"]:
            fr = frames[int(fi)]
            for opts in refodo.OPTION_SETS:
                key = f"f{int(fi)}/{opts[0]}"
                if key + "/trans" not in z.files:
                    continue
                tr, rot, st, err = refodo.track_once(orc.Odometry, cam, W, H, fr, opts)
                assert refpin.bits_equal(tr, z[key + "/trans"]) and refpin.bits_equal(rot, z[key + "/rot"]), f"{key}: pose differs from the reference class"
                assert np.array_equal(st["lastA"], z[key + "/lastA"]) and np.array_equal(st["lastb"], z[key + "/lastb"]), f"{key}: last normal equations"
                rs = z[key + "/stats"]
                icp, rgb, so3 = not opts[1] and opts[2] > 0, opts[1] or opts[2] < 100, opts[5]
                mine = np.array([st["last_icp_error"], st["last_icp_count"], st["last_rgb_error"], st["last_rgb_count"], st["last_so3_error"], st["last_so3_count"]], np.float32)
                which = ([0, 1] if icp else []) + ([2, 3] if rgb else []) + ([4, 5] if so3 else [])   # (a member the call does not write keeps its constructor value in the reference)
                assert refpin.bits_equal(mine[which], rs[which]), f"{key}: statistics {mine} vs {rs}"
                if icp:
                    es = z[key + "/err_sum_max"]
                    assert float(err.astype(np.float64).sum()) == float(es[0]) and float(err.max()) == float(es[1]), f"{key}: ICP error surface"
                checked += 1
    finally:
        orc.set_icp_arith("product")
    assert checked >= 4


def test_reference_order_trajectory_equals_the_reference_tracker():
    """VERDICT r5 item 1 (c) on the CPU: the pinned frame loop with the oracle's tracker under ORC_ICP_ARITH_REFERENCE against
    tests/golden/ref_traj_v1.npz (the same loop tracked by the reference's own RGBDOdometry class): model lists, ids, SURFEL COUNTS and
    the poses of every model identical on every frame -- the three 160x128 scenarios over their whole length (static camera; motion CRF
    with spawns; ground-truth masks), the four 640x480 ones over their first CPU_FRAMES_640_EXACT frames here (their whole length on the
    MI355X: tests/test_refpin_gpu.py, and here with COFUSION_LONG_TESTS=1: 5 min each).  trajpin.exact, no tolerance."""
    import subprocess
    import sys
    import trajpin
    if not ref.available():
        pytest.skip("oracle/_ref not built (the frame loop is the reference's text)")
    z = np.load(TRAJ_GOLDEN)
    long_run = bool(os.environ.get("COFUSION_LONG_TESTS"))
    procs = {}
    for name in trajpin.scenarios():
        F = z[name + "/poses"].shape[0]
        if name.endswith("_640") and not long_run:
            F = min(F, CPU_FRAMES_640_EXACT)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); import orc, make_ref_traj_golden as g; "
                "orc.set_icp_arith('reference'); p, i, c = g.play(%r, False, n_frames=%d); np.savez(sys.argv[1], poses=p, ids=i, counts=c)"
                % (os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__)), os.path.join(os.path.dirname(__file__), "golden"), name, F))
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"trajx_{name}_{os.getpid()}.npz")
        procs[name] = (subprocess.Popen([sys.executable, "-c", code, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, OMP_NUM_THREADS="1")), out)
    models = 0
    for name, (proc, out) in procs.items():
        _, err_text = proc.communicate(timeout=2400)
        assert proc.returncode == 0, f"{name}: {err_text.decode()[-2000:]}"
        o = np.load(out); os.remove(out)
        r = trajpin.exact(name, o["poses"], o["ids"], o["counts"], z=z)
        assert r["identical"], f"{name}: not the reference tracker's trajectory: {r}"
        assert float(np.abs(o["poses"][-1, 0, :3, 3] - o["poses"][0, 0, :3, 3]).max()) > 1e-3, f"{name}: the camera did not move"
        models = max(models, int((o["ids"] >= 0).sum(axis=1).max()))
    assert models >= 3 and len(procs) >= 7


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_gn_loop_fixture_is_what_the_reference_class_produces():
    """the committed fixture is reproducible from the reference sources (spot check: the RGB-only option set, ~25 s on the emulator;
    tests/golden/make_ref_odo_golden.py regenerates all of it)"""
    import refodo
    z, cam, frames, W, H = _odo_inputs()
    fi = int(z["frames"][0])
    opts = [o for o in refodo.OPTION_SETS if o[0] == "rgb_only"][0]
    tr, rot, st, _ = refodo.track_once(refodo.RefOdometry, cam, W, H, frames[fi], opts)
    key = f"f{fi}/rgb_only"
    assert refpin.bits_equal(tr, z[key + "/trans"]) and refpin.bits_equal(rot, z[key + "/rot"])
    assert st["last_rgb_count"] == z[key + "/stats"][3] and st["last_so3_count"] == z[key + "/stats"][5]
    assert np.array_equal(st["lastA"], z[key + "/lastA"]) and np.array_equal(st["lastb"], z[key + "/lastb"])


# ---- the frame loop: oracle vs the reference's own CoFusion::processFrame text ----------------------------------------------------
CF_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_cofusion_v1.json")


def _cf_golden():
    import json
    with open(CF_GOLDEN) as f:
        return json.load(f)["scenarios"]


def test_frame_loop_matches_reference_process_frame():
    """the oracle's restatement of the frame loop (orc_multi.MultiPipeline / orc_pipeline.StaticPipeline, which the
    C++ facade is tested against bit for bit) against CoFusion::processFrame + performSegmentation / predict / requiresFillIn /
    spawnObjectModel / moveNewModelToList / inactivateModel / getNextModelID as they stand in /root/reference/Core/CoFusion.cpp, with
    the reference's own Core/Segmentation: model list, ids, poses, surfel buffers, unseen counters, label masks and the clock identical
    in every frame (3 + 7 spawns, 2 + 5 deactivations, id re-use, ground-truth masks, fill-in tracking, single-model mode)."""
    import cfpin
    gold = _cf_golden()
    assert set(gold) == set(cfpin.SCENARIOS)
    spawns = drops = 0
    for name, ref_rows in gold.items():
        if name in cfpin.FACADE_ONLY:  # tracking switches the oracle's Python loop does not have: checked against the facade on the GPU
            continue
        orc_rows = cfpin.run_oracle(name)
        diffs = cfpin.differences(ref_rows, orc_rows)
        assert not diffs, f"{name}: " + "; ".join(diffs[:4])
        n = [len(r["ids"]) for r in ref_rows]
        spawns += sum(1 for a, b in zip(n, n[1:]) if b > a); drops += sum(1 for a, b in zip(n, n[1:]) if b < a)
    assert spawns >= 10 and drops >= 5, "the scenarios no longer exercise spawning / deactivation"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_frame_loop_fixture_is_what_the_reference_text_produces():
    """the committed fixture is reproducible from the reference sources (two scenarios, each in a process of its own)"""
    import cfpin
    gold = _cf_golden()
    for name in ("crf_two_objects", "gt_masks_three_objects"):
        rows = cfpin.run_reference_isolated(name)
        assert rows == gold[name], f"{name}: the reference frame loop no longer produces the committed fixture"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_model_argument_plumbing_runs_on_the_reference_text_and_the_pin_has_teeth():
    """Model::initICP / performTracking / fuse / clean (Core/Model/Model.cpp:350-389, 408-697) are compiled as TEXT behind a recording
    OpenGL (oracle/ref_shim/stub/glpin.h): which uniform gets which value, which texture sits on which sampler, which buffer is read and
    which written is decided by that text, and the draw handler runs the oracle's pass with what was recorded.  The frame-loop fixture
    (which the C++ facade reproduces bit for bit, tests/test_configs_gpu.py) comes out of that path -- the test above -- and stops coming
    out of it when the recorded state is corrupted the way a plumbing mistake would: depth textures on each other's units, the clean
    pass's depth sampler on the wrong unit, a time uniform one frame late."""
    import refcofusion
    import cfpin
    gold = _cf_golden()
    cam, frames = cfpin.frames_of("static")
    cf = refcofusion.RefCoFusion(cam, conf_global=10.0, spawn_offset=20, multi=False)
    before = cf.gl_draws()
    for t, (d, rgb, _, _) in enumerate(frames[:3]):
        cf.process_frame(d, rgb, timestamp=t)
    after = cf.gl_draws()
    # frame 0 initialises the map (no fuse / clean); each later frame: one data + one update draw in fuse, two feedback draws in clean
    assert tuple(a - b for a, b in zip(after, before)) == (2, 2, 4)
    for mutation in (1, 2, 3):
        try:
            rows = cfpin.run_reference_isolated("crf_two_objects", glpin_mutation=mutation)
        except subprocess.CalledProcessError:
            continue   # the handler's own cross-checks (e.g. both passes of fuse must name the same time) stopped the run: noticed
        assert rows != gold["crf_two_objects"], f"corruption {mutation} of the recorded GL state went unnoticed"


# ---- trajectory level: the reference's own arithmetic as the tracker of the frame loop ----------------------------
TRAJ_GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_traj_v1.npz")
# BASELINE.json: "pose trajectory within 1e-3 m ATE of the reference".  The reference reduces 29 f32 values per pixel in launch-shape
# dependent trees, 57 times per tracked model and frame, and solves in f32/f64 Eigen; the oracle (= the HIP path, bit for bit) sums exact
# integers.  Per call the poses differ by ~1e-6 (ODO_POSE_TOL); over a trajectory the differences feed back through the fused map.
ATE_TOL_M = 1e-3


def _ate(a, b):
    e = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=1)
    return float(np.sqrt(np.mean(e ** 2))), float(e.max())


# CPU budget: the oracle needs ~2 s per 640x480 model-frame; the long 640x480 scenarios are played for this many frames here (their full
# length on the MI355X, tests/test_configs_gpu.py, and here with COFUSION_LONG_TESTS=1).  36 since round 6 (60 until then: five minutes of an
# almost ten-minute CPU suite) -- the comparison that must not lose a frame, exact parity under the reference-order arithmetic, has its own tests
CPU_FRAMES_640 = 36
CPU_FRAMES_640_EXACT = 10   # ... and of the bit-for-bit comparison under the reference-order arithmetic (a tree-ordered f32 sum is serial in the oracle)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_trajectory_within_1mm_ate_of_the_reference_arithmetic():
    """tests/golden/ref_traj_v1.npz holds the poses / ids / surfel counts of the pinned frame loop (the text of CoFusion::processFrame) when
    every model is tracked by the reference's OWN RGBDOdometry class -- its CUDA kernels under the emulator, f32 tree reductions,
    Eigen-style solve (tests/golden/make_ref_traj_golden.py): at 160x128 40 static frames, 24 frames with two objects and the motion CRF,
    32 frames with two objects and ground-truth masks; at BASELINE.json's 640x480 100 static frames and 60 frames of each two-object
    scenario.  Here the same loops run with the oracle's exact-integer tracker (whose bits the HIP path reproduces:
    tests/test_configs_gpu.py) and tests/trajpin.compare holds them against the fixture: camera ATE <= 1e-3 m, model lists, the SURFEL
    COUNTS within a stated bound (reported: first differing frame, largest difference), every object the reference keeps for >= 10
    frames within a stated bound on every frame of its life."""
    import subprocess
    import sys
    import trajpin
    z = np.load(TRAJ_GOLDEN)
    names = trajpin.scenarios(exact_only=False)
    assert len(names) >= 7, "empty fixture"
    long_run = bool(os.environ.get("COFUSION_LONG_TESTS"))
    # one process per scenario (function-static state in Core/Segmentation, see cfpin.run_reference_isolated), all of them side by side
    procs = {}
    for name in names:
        F = z[name + "/poses"].shape[0]
        if F > CPU_FRAMES_640 and not long_run:
            F = CPU_FRAMES_640
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); import make_ref_traj_golden as g; "
                "p, i, c = g.play(%r, False, n_frames=%d); np.savez(sys.argv[1], poses=p, ids=i, counts=c)"
                % (os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__)), os.path.join(os.path.dirname(__file__), "golden"), name, F))
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"traj_{name}_{os.getpid()}.npz")
        env = dict(os.environ, OMP_NUM_THREADS="2")
        procs[name] = (subprocess.Popen([sys.executable, "-c", code, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env), out, F)
    reports = {}
    for name in names:
        proc, out, F = procs[name]
        _, err_text = proc.communicate(timeout=2400)
        assert proc.returncode == 0, f"{name}: {err_text.decode()[-2000:]}"
        o = np.load(out); os.remove(out)
        reports[name] = trajpin.compare(name, o["poses"], o["ids"], o["counts"], z=z)
    assert min(reports[n]["frames"] for n in ("static_camera_640", "crf_two_objects_640", "gt_masks_two_objects_640")) >= CPU_FRAMES_640
    # round 5: two textured boxes with ground-truth masks -- lists identical throughout, both objects compared over their whole life, counts
    # within 2 %, the large one (11-21 k surfels) within the tight bound
    boxes = reports["gt_masks_two_boxes_640"]
    assert boxes["lists_identical_frames"] == boxes["frames"] >= CPU_FRAMES_640 and len(boxes["objects"]) == 2
    assert all(o["frames"] >= CPU_FRAMES_640 - 10 and o["count_max_rel_diff"] <= 0.02 for o in boxes["objects"].values()), boxes["objects"]
    assert any(o["stable_in_reference"] and o["max_m"] <= trajpin.OBJECT_BOUND_M for o in boxes["objects"].values()), boxes["objects"]
    assert sum(len(r["objects"]) for r in reports.values()) >= 4, "no object trajectory was compared"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_trajectory_fixture_is_what_the_reference_tracker_produces():
    """the committed trajectory fixture is reproducible from the reference sources: the first TRACKED frames of the static scenario through
    the reference's own RGBDOdometry class under the emulator (~9 s per frame; tests/golden/make_ref_traj_golden.py regenerates all of it)"""
    import subprocess
    import sys
    z = np.load(TRAJ_GOLDEN)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r); import make_ref_traj_golden as g; "
            "p, i, c = g.play('static_camera', True, n_frames=4); np.savez(sys.argv[1], poses=p, counts=c)"
            % (os.path.dirname(__file__), os.path.dirname(os.path.dirname(__file__)), os.path.join(os.path.dirname(__file__), "golden")))
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"traj_live_{os.getpid()}.npz")
    subprocess.run([sys.executable, "-c", code, out], check=True, capture_output=True)
    o = np.load(out); os.remove(out)
    for t in (1, 2, 3):
        assert refpin.bits_equal(o["poses"][t, 0], z["static_camera/poses"][t, 0]), f"frame {t}: pose of the reference-tracked run"
        assert int(o["counts"][t, 0]) == int(z["static_camera/counts"][t, 0])
    assert float(np.abs(o["poses"][1, 0, :3, 3]).max()) > 5e-3, "degenerate: the camera did not move"


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_fusion_weight_against_the_reference_text():
    """VERDICT r4 (missing #3): Model::computeFusionWeight and Model::rodrigues2 (Model.cpp:391-406, 817-865) were restated, not pinned.
    build_ref.py now cuts their TEXT out of the reference and compiles it (oracle/ref_shim/ref_weight.cpp; Eigen's fixed-size matrices and
    `inverse()` by the stand-in's stated conventions, JacobiSVD as the identity re-orthonormalisation our restatements state too).  The
    oracle's orc_fusion_weight and the library's cf_fusion_weight (host arithmetic, callable without a GPU) are held against it over 3000
    pose pairs: frame-to-frame motions from 0 to beyond the 1 cm / 0.01 rad saturation, exact identities (the s < 1e-5 branch with
    c > 0), half turns (c <= 0).  Oracle and library agree BIT FOR BIT; against the reference text they agree to 1e-4 of the weight (observed
    3.6e-5) -- getLastTransform is a general 4x4 inverse times a matrix there (Model.h:216) and a rigid inverse here: with poses up to
    2 m from the origin that moves |t| by a few 1e-7 m, and the weight divides it by 0.01."""
    import ctypes as C
    from co_fusion_amd import lib as cflib
    import orc_pipeline as op
    L = ref.lib()
    L.ref_fusion_weight.restype = C.c_float
    H = cflib.load()
    H.cf_fusion_weight.restype = C.c_float
    P = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    rng = np.random.default_rng(11)

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

    def pose(R, t):
        T = np.eye(4, dtype=np.float32); T[:3, :3] = R; T[:3, 3] = t
        return T
    worst, seen = 0.0, set()
    for k in range(3000):
        base = pose(rot(rng.normal(size=3), rng.uniform(0, 3.0)), rng.uniform(-2, 2, size=3))
        kind = k % 6
        if kind == 0:
            d = pose(np.eye(3), np.zeros(3))                                  # no motion: s < 1e-5, c > 0
        elif kind == 1:
            d = pose(rot(rng.normal(size=3), np.pi), rng.uniform(-1e-3, 1e-3, size=3))   # half turn: s < 1e-5, c <= 0
        elif kind == 2:
            d = pose(rot(rng.normal(size=3), rng.uniform(0, 0.004)), rng.uniform(-0.004, 0.004, size=3))
        elif kind == 3:
            d = pose(rot(rng.normal(size=3), rng.uniform(0, 0.03)), rng.uniform(-0.02, 0.02, size=3))   # around and beyond the saturation
        elif kind == 4:
            d = pose(rot(rng.normal(size=3), rng.uniform(0, 1e-6)), rng.uniform(-1e-6, 1e-6, size=3))
        else:
            d = pose(rot(rng.normal(size=3), rng.uniform(0, 0.5)), rng.uniform(-0.3, 0.3, size=3))
        last = (base.astype(np.float64) @ d.astype(np.float64)).astype(np.float32)
        mult = np.float32(rng.choice([1.0, 100.0, 0.5]))
        a = np.ascontiguousarray(base.reshape(16)); b = np.ascontiguousarray(last.reshape(16))
        w_ref = float(L.ref_fusion_weight(P(a), P(b), C.c_float(mult)))
        w_orc = op.fusion_weight(base, last, float(mult))
        w_lib = float(H.cf_fusion_weight(P(a), P(b), C.c_float(mult)))
        assert np.float32(w_orc).tobytes() == np.float32(w_lib).tobytes(), f"pair {k}: oracle {w_orc} vs library {w_lib}"
        worst = max(worst, abs(w_orc - w_ref) / float(mult))
        seen.add(round(w_ref / float(mult), 1))
    assert worst <= 1e-4, f"fusion weight differs from the reference text by {worst} (per unit multiplier)"
    assert 0.5 in seen and 1.0 in seen and len(seen) >= 4, f"the pairs did not cover the weight's range: {sorted(seen)}"
