"""Shared scenario of the segmentation pin: the inputs Segmentation::performSegmentationCRF sees over the first frames of a seeded
multi-object run (captured from the oracle pipeline), the oracle's answer and -- where oracle/_ref is built -- the answer of the
reference's OWN Core/Segmentation sources (ref_segment_crf, oracle/ref_shim/ref_seg.cpp).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import hashlib
import warnings

import numpy as np

import orc
import orc_multi as om

W, H = 320, 240
N_OBJ, FRAMES = 3, 9
STAT_RTOL = 2e-5   # per-superpixel means are f32 running sums in the reference, exact fixed-point sums here (Slic.h:61-66)


class RefSegParams(C.Structure):
    """ref_seg_params: the three pairwise SIGMAS where orc_seg_params holds their f32 reciprocals (the reference's setters divide)"""
    _fields_ = [(n, C.c_float) for n in ("unaryWeightError", "unaryKError", "unaryThresholdNew", "weightAppearance", "weightSmoothness",
                                         "sigmaRGB", "sigmaDepth", "sigmaPos", "minRelSizeNew", "maxRelSizeNew")] + [("crfIterations", C.c_int)]

    @staticmethod
    def defaults():
        return RefSegParams(75.0, 0.0375, 5.5, 7.0, 2.0, 10.0, 0.9, 1.8, 0.015, 0.4, 10)


def capture():
    """Run the oracle pipeline; returns the list of performSegmentationCRF calls: inputs + the oracle's result."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    from co_fusion_amd import synth
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=N_OBJ)
    pipe = om.MultiPipeline(cam, conf_global=0.5, spawn_offset=2)
    calls = []
    orig = om.segment_crf

    def hook(params, rgba, depth, ids, icps, vcs, next_id, allow_new):
        res = orig(params, rgba, depth, ids, icps, vcs, next_id, allow_new)
        calls.append(dict(rgba=rgba.copy(), depth=np.array(depth, np.float32), ids=list(ids), icps=[np.array(a, np.float32) for a in icps],
                          vcs=[np.array(a, np.float32) for a in vcs], next_id=int(next_id), allow_new=bool(allow_new), oracle=res))
        return res

    om.segment_crf = hook
    try:
        for t in range(FRAMES):
            d, rgb, _, _ = sc.render(cam, t, noise=True)
            pipe.process_frame(d, synth.rgb_to_rgba(rgb))
    finally:
        om.segment_crf = orig
    return calls


def input_digest(c):
    h = hashlib.sha256()
    for a in [c["rgba"], c["depth"], *c["icps"], *c["vcs"]]:
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(repr((c["ids"], c["next_id"], c["allow_new"])).encode())
    return h.hexdigest()


def run_reference(c):
    """the reference's performSegmentationCRF on one captured call -> dict like orc_multi.segment_crf's"""
    import ref
    L = ref.lib()
    n = len(c["ids"]); K = (W // 16) * (H // 16)
    ids = (C.c_uint * n)(*c["ids"])
    icp_keep = [orc.f32(a) for a in c["icps"]]; vc_keep = [orc.f32(a) for a in c["vcs"]]
    icp_arr = (C.c_void_p * n)(*[a.ctypes.data for a in icp_keep]); vc_arr = (C.c_void_p * n)(*[a.ctypes.data for a in vc_keep])
    full = np.zeros((H, W), np.uint8); models = (om.SegModel * (n + 1))(); n_out = C.c_int(); has_new = C.c_int(); rng = C.c_float()
    low_d = np.zeros(K, np.float32)
    rgb3 = np.ascontiguousarray(c["rgba"][..., :3])
    rp = RefSegParams.defaults()
    L.ref_segment_crf(C.byref(rp), W, H, orc.P(rgb3), orc.P(orc.f32(c["depth"])), n, ids, icp_arr, vc_arr, C.c_uint(c["next_id"]),
                      int(c["allow_new"]), orc.P(full), models, C.byref(n_out), C.byref(has_new), C.byref(rng), orc.P(low_d), None, None)
    md = [dict(id=m.id, superPixelCount=m.superPixelCount, avgConfidence=m.avgConfidence, depthMean=m.depthMean, depthStd=m.depthStd,
               top=m.top, right=m.right, bottom=m.bottom, left=m.left) for m in models[:n_out.value]]
    return dict(full=full, modelData=md, hasNewLabel=bool(has_new.value), depthRange=rng.value, lowDepth=low_d)


INT_FIELDS = ("id", "superPixelCount", "top", "right", "bottom", "left")
FLOAT_FIELDS = ("avgConfidence", "depthMean", "depthStd")


def pack_result(r):
    """flat arrays of one result (for the committed fixture)"""
    md = r["modelData"]
    return dict(full=np.packbits(r["full"] != 0), full_labels=np.unique(r["full"]), full_sha=hashlib.sha256(r["full"].tobytes()).hexdigest(),
                ints=np.array([[m[f] for f in INT_FIELDS] for m in md], np.int64).reshape(len(md), len(INT_FIELDS)),
                floats=np.array([[m[f] for f in FLOAT_FIELDS] for m in md], np.float32).reshape(len(md), len(FLOAT_FIELDS)),
                has_new=np.array([int(r["hasNewLabel"])]), depth_range=np.array([r["depthRange"]], np.float32))


def compare(oracle_res, ref_packed, what):
    """the oracle against the reference's answer: labels, counts, boxes, decisions identical; float statistics to STAT_RTOL"""
    got = pack_result(oracle_res)
    assert got["full_sha"] == str(ref_packed["full_sha"]), f"{what}: full-resolution label mask differs from the reference's"
    assert np.array_equal(got["ints"], ref_packed["ints"]), f"{what}: ids / super-pixel counts / bounding boxes\n{got['ints']}\n{ref_packed['ints']}"
    assert int(got["has_new"][0]) == int(ref_packed["has_new"][0]), f"{what}: hasNewLabel"
    np.testing.assert_allclose(got["floats"], ref_packed["floats"], rtol=STAT_RTOL, atol=1e-6, err_msg=f"{what}: avgConfidence / depthMean / depthStd")
    np.testing.assert_allclose(got["depth_range"], ref_packed["depth_range"], rtol=STAT_RTOL, err_msg=f"{what}: depthRange")
