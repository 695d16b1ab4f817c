"""GPU parity of the device-resident segmentation stage (cf_seg_slic -> cf_seg_sums -> cf_seg_infer -> cf_seg_fetch) against the
oracle's Segmentation::performSegmentationCRF on label images the frame loop never produces: checkerboards (one component per
superpixel: more components than the LDS statistics hold, up to 64 components per wave), one-superpixel stripes, random labels,
empty models, invalid depth -- the connected-components / gate / statistics code paths behind the smooth masks of the real run."""
import ctypes as C
import warnings

import numpy as np
import pytest

import orc_multi as om

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", category=RuntimeWarning)

W, H = 640, 480
GX, GY = W // 16, H // 16


class SegModel(C.Structure):
    _fields_ = [("id", C.c_uint32), ("superPixelCount", C.c_uint32), ("avgConfidence", C.c_float), ("depthMean", C.c_float),
                ("depthStd", C.c_float), ("top", C.c_int32), ("right", C.c_int32), ("bottom", C.c_int32), ("left", C.c_int32)]


class SegResult(C.Structure):
    _fields_ = [("has_new_label", C.c_int32), ("n_models", C.c_int32), ("depth_range", C.c_float), ("model", SegModel * 256)]


def _bits(x):
    return np.float32(x).view(np.uint32)


def _run_hip(ctx, seg, params, rgba, depth, ids, icp, vc, next_id, allow_new):
    n = len(ids)
    keep = [ctx.to_device(rgba), ctx.to_device(depth)] + [ctx.to_device(a) for a in icp] + [ctx.to_device(a) for a in vc]
    t_rgba, t_depth = keep[0], keep[1]
    icp_arr = (C.c_void_p * n)(*[t.data_ptr() for t in keep[2:2 + n]])
    vc_arr = (C.c_void_p * n)(*[t.data_ptr() for t in keep[2 + n:]])
    full = ctx.to_device(np.zeros((H, W), np.uint8))
    ctx._check(ctx.lib.cf_seg_slic(seg, C.c_void_p(t_rgba.data_ptr())))
    sums = C.c_void_p(); words = C.c_uint64()
    ctx._check(ctx.lib.cf_seg_sums(seg, C.c_void_p(t_depth.data_ptr()), n, icp_arr, vc_arr, C.byref(sums), C.byref(words)))
    id_arr = (C.c_uint32 * n)(*ids)
    ctx._check(ctx.lib.cf_seg_infer(seg, C.byref(params), C.c_void_p(t_rgba.data_ptr()), n, id_arr, C.c_uint32(next_id), int(allow_new),
                                    C.c_void_p(full.data_ptr())))
    res = SegResult()
    low = np.zeros(GX * GY, np.uint8)
    ctx._check(ctx.lib.cf_seg_fetch(seg, C.byref(res), low.ctypes.data_as(C.c_void_p)))
    return res, low.reshape(GY, GX), full.cpu().numpy()


_REFS = {}


def _ref(sc):  # the oracle's result of a scenario, shared by the tests of this module
    if sc[0] not in _REFS:
        _REFS[sc[0]] = om.segment_crf(sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7], sc[8])
    return _REFS[sc[0]]


def _compare(name, res, low, full, ref):
    assert np.array_equal(low, ref["low"]), f"{name}: low-resolution labels differ in {np.count_nonzero(low != ref['low'])} superpixels"
    assert np.array_equal(full, ref["full"]), f"{name}: full-resolution mask"
    assert bool(res.has_new_label) == ref["hasNewLabel"], f"{name}: hasNewLabel"
    assert _bits(res.depth_range) == _bits(ref["depthRange"]), f"{name}: depth range"
    assert res.n_models == len(ref["modelData"]), f"{name}: rows {res.n_models} vs {len(ref['modelData'])}"
    for i, r in enumerate(ref["modelData"]):
        g = res.model[i]
        assert (g.id, g.superPixelCount) == (r["id"], r["superPixelCount"]), f"{name} row {i}: id / count"
        assert (g.top, g.right, g.bottom, g.left) == (r["top"], r["right"], r["bottom"], r["left"]), f"{name} row {i}: box"
        for k in ("avgConfidence", "depthMean", "depthStd"):
            a, b = getattr(g, k), r[k]
            assert _bits(a) == _bits(b) or (a != a and b != b), f"{name} row {i}: {k} {a} vs {b}"


def _block_image(values):
    """[GY, GX] per-superpixel values -> [H, W] image constant over each 16x16 block"""
    return np.repeat(np.repeat(np.asarray(values, np.float32), 16, 0), 16, 1)


def _scenarios():
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:GY, 0:GX]
    rgba_flat = np.zeros((H, W, 4), np.uint8); rgba_flat[..., :3] = 120; rgba_flat[..., 3] = 255
    rgba_noise = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8); rgba_noise[..., 3] = 255
    depth_ramp = (1.0 + 2.0 * np.arange(W, dtype=np.float32)[None, :] / W + 0.5 * np.arange(H, dtype=np.float32)[:, None] / H).astype(np.float32)

    def vc(conf, depth):
        v = np.zeros((H, W, 4), np.float32)
        v[..., 2] = depth; v[..., 3] = conf
        return v

    def errs_from_labels(lab, n, lo=0.0005, hi=0.2):
        return [_block_image(np.where(lab == m, lo, hi)) for m in range(n)]

    unary_only = om.SegParams.defaults()
    unary_only.weightAppearance = 0.0; unary_only.weightSmoothness = 0.0
    out = []
    # one component per superpixel: 1200 components (beyond the LDS statistics), 64 different components in every wave
    lab = (yy + xx) % 2
    out.append(("checkerboard of two models", unary_only, rgba_flat, depth_ramp, [0, 3], errs_from_labels(lab, 2), [vc(1.0, depth_ramp)] * 2, 4, False))
    # one-superpixel-wide stripes of three models: tall thin components, 13-14 per label
    lab = xx % 3
    out.append(("vertical stripes of three models", unary_only, rgba_noise, depth_ramp, [0, 1, 2], errs_from_labels(lab, 3), [vc(1.0, depth_ramp)] * 3, 3, False))
    # random labels of five models, a new label allowed: hundreds of components, keep-largest ties, size gates of the new label
    lab = rng.integers(0, 5, size=(GY, GX))
    e = errs_from_labels(lab, 5, hi=0.08)
    e[0][:] = np.where(_block_image(rng.random((GY, GX)) < 0.15) > 0, 0.3, e[0])  # some superpixels nobody explains -> new label
    for m in range(1, 5):
        e[m][:] = np.where(e[0] == 0.3, 0.3, e[m])
    out.append(("random labels of five models + new label", unary_only, rgba_noise, depth_ramp, [0, 2, 5, 7, 9], e, [vc(1.0, depth_ramp)] * 5, 10, True))
    # the default CRF on the same inputs (smoothing merges the noise), a model that gets nothing, low-confidence regions
    conf = _block_image(np.where(xx < GX // 3, 0.2, 1.0))
    out.append(("default CRF, starved model, low confidence", om.SegParams.defaults(), rgba_noise, depth_ramp, [0, 2, 5, 7, 9], e,
                [vc(conf, depth_ramp)] * 4 + [vc(0.0, depth_ramp)], 10, True))
    # superpixels without valid depth (holes) and an object hugging the border (border gate)
    d = depth_ramp.copy()
    d[:64, :] = 0.0; d[200:264, 300:364] = 0.0
    lab = np.zeros((GY, GX), np.int64); lab[:, :1] = 1; lab[10:20, 15:25] = 2
    out.append(("depth holes + border object", om.SegParams.defaults(), rgba_noise, d, [0, 1, 2], errs_from_labels(lab, 3), [vc(1.0, d)] * 3, 3, True))
    # MORE THAN 16 LABELS (round 4: the label dimension follows the context's max_models up to the reference's 255 ids): 24 models in
    # vertical bands + a new label, unary only and with the default CRF; 40 models, the accumulation in three tiles of 16
    for n, params, name in ((24, unary_only, "24 models + new label, unary only"), (24, om.SegParams.defaults(), "24 models + new label, default CRF"),
                            (40, unary_only, "40 models")):
        lab = (xx * n // GX + (yy // 5) * 3) % n
        e = errs_from_labels(lab, n, hi=0.08)
        hole = _block_image(rng.random((GY, GX)) < 0.1) > 0
        for m in range(n):
            e[m][:] = np.where(hole, 0.3, e[m])
        out.append((name, params, rgba_noise, depth_ramp, [0] + list(range(3, 3 + n - 1)), e, [vc(1.0, depth_ramp)] * n, 3 + n - 1, n == 24))
    return out


def test_segmentation_stage_on_adversarial_label_images():
    from co_fusion_amd import api, synth
    cam = synth.Camera.scaled(W, H)
    ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_models=48)
    seg = C.c_void_p()
    ctx._check(ctx.lib.cf_seg_create(ctx.h, C.byref(seg)))
    components = []
    for sc in _scenarios():
        name, params, rgba, depth, ids, icp, vcs, next_id, allow_new = sc
        ref = _ref(sc)
        p = params.__class__.from_buffer_copy(params)
        for rep in range(2):  # twice: the kernels leave their accumulators clean for the next frame
            res, low, full = _run_hip(ctx, seg, p, rgba, depth, ids, icp, vcs, next_id, allow_new)
            _compare(f"{name} (pass {rep})", res, low, full, ref)
        components.append(len(np.unique(ref["low"])))
    assert max(components) >= 2
    ctx.lib.cf_seg_destroy(seg)
    ctx.close()


class SegJob(C.Structure):
    _fields_ = [("seg", C.c_void_p), ("depth", C.c_void_p), ("n_models", C.c_int32), ("icp_err", C.c_void_p), ("vertconf4", C.c_void_p),
                ("rgba", C.c_void_p), ("model_ids", C.c_void_p), ("next_model_id", C.c_uint32), ("allow_new", C.c_int32), ("full_dev", C.c_void_p)]


def test_batched_segmentation_of_several_segmenters_matches_the_oracle():
    """cf_seg_run_batch: the chains of several segmenters of one context (the sequences of a lock-step group) through shared launches --
    every scenario on a segmenter of its own, all of one parameter set in ONE call (more than eight: two chunks; a scenario with more
    than 16 labels: the per-segmenter fallback), twice (clean accumulators), each against the oracle."""
    from co_fusion_amd import api, synth
    cam = synth.Camera.scaled(W, H)
    ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_models=48)
    ctx.lib.cf_seg_run_batch.restype = C.c_int
    ctx.lib.cf_seg_run_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    groups = {}
    for sc in _scenarios():
        groups.setdefault(bytes(sc[1]), []).append(sc)
    sizes = []
    for key, scs in groups.items():
        small = [sc for sc in scs if len(sc[4]) + (1 if sc[8] else 0) <= 16]
        # all-small: batched launches; with a big one: the per-segmenter fallback; the small ones three times over: more than eight
        # segmenters in one call (two chunks of the batched launches)
        batches = ([small, scs] if len(small) != len(scs) else [scs]) + ([small * 3] if 2 < len(small) and len(small) * 3 > 8 else [])
        for batch in batches:
            sizes.append(len(batch))
            segs, keep, jobs = [], [], (SegJob * len(batch))()
            for k, (name, params, rgba, depth, ids, icp, vcs, next_id, allow_new) in enumerate(batch):
                seg = C.c_void_p()
                ctx._check(ctx.lib.cf_seg_create(ctx.h, C.byref(seg)))
                segs.append(seg)
                n = len(ids)
                t = [ctx.to_device(rgba), ctx.to_device(depth), ctx.to_device(np.zeros((H, W), np.uint8))] + [ctx.to_device(a) for a in icp] + [ctx.to_device(a) for a in vcs]
                icp_arr = (C.c_void_p * n)(*[x.data_ptr() for x in t[3:3 + n]])
                vc_arr = (C.c_void_p * n)(*[x.data_ptr() for x in t[3 + n:]])
                id_arr = (C.c_uint32 * n)(*ids)
                keep.append((t, icp_arr, vc_arr, id_arr))
                jobs[k] = SegJob(seg.value, t[1].data_ptr(), n, C.cast(icp_arr, C.c_void_p).value, C.cast(vc_arr, C.c_void_p).value,
                                 t[0].data_ptr(), C.cast(id_arr, C.c_void_p).value, next_id, int(allow_new), t[2].data_ptr())
            p = batch[0][1].__class__.from_buffer_copy(batch[0][1])
            refs = [_ref(sc) for sc in batch]
            for rep in range(2):
                for k in range(len(batch)):
                    ctx._check(ctx.lib.cf_seg_slic(segs[k], C.c_void_p(keep[k][0][0].data_ptr())))
                ctx._check(ctx.lib.cf_seg_run_batch(ctx.h, C.byref(p), C.cast(jobs, C.c_void_p), len(batch)))
                for k, sc in enumerate(batch):
                    res = SegResult()
                    low = np.zeros(GX * GY, np.uint8)
                    ctx._check(ctx.lib.cf_seg_fetch(segs[k], C.byref(res), low.ctypes.data_as(C.c_void_p)))
                    _compare(f"batched {sc[0]} (pass {rep}, {len(batch)} jobs)", res, low.reshape(GY, GX), keep[k][0][2].cpu().numpy(), refs[k])
            for seg in segs:
                ctx.lib.cf_seg_destroy(seg)
    assert max(sizes) >= 2
    ctx.close()


def test_segmentation_stage_with_a_ragged_superpixel_count(monkeypatch):
    """176 x 144: 11 x 9 = 99 superpixels, not a multiple of sixteen -- the blocked sequential sums of the statistics pad their last block
    (seg_post_kernel) or fall back to the strided flavour (seg_unary_kernel's average confidences); and a single wave's worth of superpixels
    per row of lanes instead of 1200 against 1024."""
    import sys
    from co_fusion_amd import api, synth
    me = sys.modules[__name__]
    w, h = 176, 144
    for k, v in (("W", w), ("H", h), ("GX", w // 16), ("GY", h // 16)):
        monkeypatch.setattr(me, k, v)
    monkeypatch.setattr(me, "_REFS", {})
    cam = synth.Camera.scaled(w, h)
    ctx = api.Context(w, h, cam.fx, cam.fy, cam.cx, cam.cy, max_models=48)
    seg = C.c_void_p()
    ctx._check(ctx.lib.cf_seg_create(ctx.h, C.byref(seg)))
    ran = 0
    for sc in _scenarios()[:6]:
        name, params, rgba, depth, ids, icp, vcs, next_id, allow_new = sc
        ref = _ref(sc)
        p = params.__class__.from_buffer_copy(params)
        for rep in range(2):
            res, low, full = _run_hip(ctx, seg, p, rgba, depth, ids, icp, vcs, next_id, allow_new)
            _compare(f"{name} at {w}x{h} (pass {rep})", res, low, full, ref)
        ran += 1
    assert ran == 6 and (me.GX * me.GY) % 16 != 0
    ctx.lib.cf_seg_destroy(seg)
    ctx.close()
