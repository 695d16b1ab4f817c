"""GPU parity of the C++ facade (libcofusion.so: CoFusion / Model / Segmentation over the C-ABI) against
the oracle's restatement of CoFusion::processFrame -- free running, no re-synchronisation: poses, surfel
buffers, counts and label masks must stay identical frame after frame."""
import ctypes as C
import warnings

import numpy as np
import pytest

import orc
import orc_multi as om
import orc_pipeline as op
from co_fusion_amd import synth

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", category=RuntimeWarning)

W, H = 320, 240


def _same(a, b, what):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ok = a.view(np.uint8) == b.view(np.uint8)
    if not ok.all():
        if a.dtype.kind == "f":
            ok2 = (a == b) | ((a != a) & (b != b))
            assert ok2.all(), f"{what}: {np.count_nonzero(~ok2)} of {ok2.size} values differ"
        else:
            assert False, f"{what}: {np.count_nonzero(~ok)} bytes differ"


def test_slic_and_crf_kernels_exact():
    """cf_seg_slic / cf_seg_crf against the oracle's SLIC and exact mean field."""
    from co_fusion_amd import api
    cam = synth.Camera.scaled(W, H)
    ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
    sc = synth.Scene(n_obj=3)
    d, rgb, _, _ = sc.render(cam, 5, noise=True)
    rgba = synth.rgb_to_rgba(rgb)
    seg = C.c_void_p()
    ctx._check(ctx.lib.cf_seg_create(ctx.h, C.byref(seg)))
    t = ctx.to_device(rgba)
    ctx._check(ctx.lib.cf_seg_slic(seg, C.c_void_p(t.data_ptr())))
    ptr = C.c_void_p(); nb = C.c_uint64()
    ctx._check(ctx.lib.cf_seg_labels(seg, C.byref(ptr), C.byref(nb)))
    lab = np.empty((H, W), np.int32)
    ctx._check(ctx.lib.cf_memcpy_d2h(ctx.h, lab.ctypes.data_as(C.c_void_p), ptr, nb))
    ref = om.slic(rgba)
    _same(lab, ref, "SLIC labels")
    assert len(np.unique(ref)) > 250

    K = (W // 16) * (H // 16)
    rng = np.random.default_rng(3)
    L = 4
    unary = rng.uniform(0.0, 6.0, size=(K, L)).astype(np.float32)
    f1 = np.stack([(np.arange(K) % (W // 16)) / 2.0, (np.arange(K) // (W // 16)) / 2.0], -1).astype(np.float32)
    f2 = rng.uniform(0, 8, size=(K, 6)).astype(np.float32)
    Q = np.zeros((K, L), np.float32)
    Qr = np.zeros((K, L), np.float32)
    ctx._check(ctx.lib.cf_seg_crf(seg, unary.ctypes.data_as(C.c_void_p), L, f1.ctypes.data_as(C.c_void_p), f2.ctypes.data_as(C.c_void_p),
                                  C.c_float(2.0), C.c_float(7.0), 10, Q.ctypes.data_as(C.c_void_p)))
    orc.lib.orc_crf_meanfield(orc.P(unary), L, K, orc.P(f1), orc.P(f2), C.c_float(2.0), C.c_float(7.0), 10, orc.P(Qr))
    _same(Q, Qr, "CRF marginals")
    np.testing.assert_allclose(Q.sum(1), 1.0, atol=1e-5)
    ctx.lib.cf_seg_destroy(seg)
    ctx.close()


def test_facade_static_matches_oracle():
    from co_fusion_amd import facade
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=0)
    ref = op.StaticPipeline(cam, conf_global=0.5)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, enable_multiple_models=0)
    for t in range(6):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        rp, rn = ref.process_frame(d, synth.rgb_to_rgba(rgb))
        cf.process_frame(d, rgb, timestamp=t)
        info = cf.model_info(0)
        assert info["count"] == rn, f"frame {t}: count"
        _same(info["pose"], rp, f"frame {t}: pose")
        _same(cf.model_download(0), ref.surfels, f"frame {t}: surfels")
    cf.close()


@pytest.mark.parametrize("use_gt_mask", [False, True])
def test_facade_multi_model_matches_oracle(use_gt_mask):
    """Moving objects, segmentation on (CRF or ground-truth masks), model spawning: free-running comparison."""
    from co_fusion_amd import facade
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=2)
    ref = om.MultiPipeline(cam, conf_global=0.5, spawn_offset=2)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, model_spawn_offset=2,
                         enable_multiple_models=1)
    spawned = False
    for t in range(8):
        d, rgb, lab, _ = sc.render(cam, t, noise=True)
        rgba = synth.rgb_to_rgba(rgb)
        gt = (lab * 40).astype(np.uint8) if use_gt_mask else None
        ref.process_frame(d, rgba, gt_mask=gt)
        cf.process_frame(d, rgb, mask=gt, timestamp=t)
        assert cf.num_models == len(ref.models), f"frame {t}: model count {cf.num_models} vs {len(ref.models)}"
        if t > 0:
            _same(cf.mask(), ref.mask, f"frame {t}: label mask")
        for i, m in enumerate(ref.models):
            info = cf.model_info(i)
            assert info["id"] == m.id
            assert info["count"] == m.surfels.shape[0], f"frame {t} model {i}: count {info['count']} vs {m.surfels.shape[0]}"
            _same(info["pose"], m.pose, f"frame {t} model {i}: pose")
            _same(np.float32(info["conf_threshold"]), np.float32(m.conf_threshold), f"frame {t} model {i}: conf threshold")
            _same(cf.model_download(i), m.surfels, f"frame {t} model {i}: surfels")
        spawned = spawned or len(ref.models) > 1
    assert spawned, "no object model was spawned: the multi-model path was not exercised"
    cf.close()


def test_klg_driven_run_and_exporters(tmp_path):
    """a .klg log drives the pipeline; savePly / exportPoses write the reference's formats."""
    from co_fusion_amd import facade, klg
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=0)
    path = tmp_path / "seq.klg"
    with klg.KlgWriter(path, W, H) as wr:
        for t in range(5):
            d, rgb, _, _ = sc.render(cam, t, noise=True)
            wr.write(33333 * t, d, rgb)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, enable_multiple_models=0,
                         enable_pose_logging=1)
    ref = op.StaticPipeline(cam, conf_global=0.5)
    for ts, d, rgb in klg.KlgReader(path, W, H):
        cf.process_frame(d, rgb, timestamp=ts)
        ref.process_frame(d, synth.rgb_to_rgba(rgb))       # the oracle sees the same mm-quantised frames
    info = cf.model_info(0)
    _same(info["pose"], ref.pose, "pose after the klg run")
    prefix = str(tmp_path) + "/"
    assert cf.export_poses(prefix) == 1 and cf.save_ply(prefix) == 1
    lines = open(prefix + "poses-0.txt").read().strip().splitlines()
    assert len(lines) == 5
    last = np.array(lines[-1].split()[1:], np.float64)
    assert int(lines[-1].split()[0]) == 33333 * 4
    np.testing.assert_allclose(last[:3], info["pose"][:3, 3], rtol=1e-5, atol=1e-7)   # background: cam -> world; six significant digits as
    # the reference's `fs << p.p(i)` writes them
    assert abs(np.linalg.norm(last[3:]) - 1.0) < 1e-5
    raw = open(prefix + "cloud-0.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    nv = int([l for l in head.decode().splitlines() if l.startswith("element vertex")][0].split()[-1])
    surf = cf.model_download(0)
    keep = surf[surf[:, 3] > info["conf_threshold"]]
    assert nv == keep.shape[0] and len(body) == nv * 31
    rec = np.frombuffer(body[:31], np.dtype([("p", "<f4", 3), ("c", "u1", 3), ("n", "<f4", 3), ("r", "<f4")]))[0]
    np.testing.assert_allclose(rec["p"], keep[0, :3], atol=1e-6)      # Tp = identity for the background model
    np.testing.assert_allclose(rec["n"], -keep[0, 8:11], atol=1e-6)
    assert rec["r"] == keep[0, 11]
    cf.close()


def _run_static_pair(frames, cam, Wx, Hx, max_surfels=1 << 20):
    from co_fusion_amd import facade
    ref = op.StaticPipeline(cam, conf_global=0.5)
    cf = facade.CoFusion(Wx, Hx, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=max_surfels, conf_global_init=0.5, enable_multiple_models=0)
    for t, (d, rgb) in enumerate(frames):
        rp, rn = ref.process_frame(d, synth.rgb_to_rgba(rgb))
        cf.process_frame(d, rgb, timestamp=t)
        info = cf.model_info(0)
        assert info["count"] == rn, f"frame {t}: count {info['count']} vs {rn}"
        _same(info["pose"], rp, f"frame {t}: pose")
        _same(cf.model_download(0), ref.surfels, f"frame {t}: surfels")
    cf.close()


def test_degenerate_frames_match_oracle():
    """Edge cases of the input domain: a frame without any valid depth, a frame with a large hole and out-of-range
    depths (< 0.3 m, > cutoff), black colour: nothing to track / fuse must behave exactly like the oracle."""
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=0)
    frames = []
    for t in range(6):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        d = d.copy(); rgb = rgb.copy()
        if t == 2:
            d[:] = 0                                   # sensor drop-out: no valid pixel at all
        if t == 3:
            d[40:160, 60:260] = 0                      # hole
            d[:30, :] = 0.1                            # closer than the bilateral gate
            d[-30:, :] = 9.0                           # beyond the depth cut-off
        if t == 4:
            rgb[:] = 0                                 # no photometric information
        frames.append((d, rgb))
    _run_static_pair(frames, cam, W, H)


def test_full_resolution_matches_oracle():
    """BASELINE.json's size (640x480), three frames, free running."""
    Wf, Hf = 640, 480
    cam = synth.Camera.scaled(Wf, Hf)
    sc = synth.Scene(n_obj=0)
    frames = [sc.render(cam, t, noise=True)[:2] for t in range(3)]
    _run_static_pair(frames, cam, Wf, Hf, max_surfels=1 << 21)


def test_surfel_buffer_overflow_is_an_error_not_a_crash():
    from co_fusion_amd import facade
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=0)
    d, rgb, _, _ = sc.render(cam, 0, noise=True)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 12, enable_multiple_models=0)   # 4096 < 320*240
    with pytest.raises(facade.CoFusionError):
        cf.process_frame(d, rgb, timestamp=0)
    cf.close()


def _read_png_gray8(path):
    import struct, zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", 0
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        if typ == b"IHDR":
            w, h, depth, colour = struct.unpack(">IIBB", body[:10])
            assert (depth, colour) == (8, 0)
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:]


def test_segmentation_png_export(tmp_path):
    """exportSegmentation (CoFusion.cpp:235-240): per segmented frame an 8-bit label PNG, rejected labels (255) as 0."""
    from co_fusion_amd import facade
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=2)
    cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, model_spawn_offset=2,
                         enable_multiple_models=1)
    prefix = str(tmp_path) + "/"
    cf.set_export_segmentation(prefix)
    for t in range(5):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        tick = cf.tick
        cf.process_frame(d, rgb, timestamp=t)
        if t > 0:
            img = _read_png_gray8(prefix + f"Segmentation{tick}.png")
            m = cf.mask().copy()
            m[m > 254] = 0
            _same(img, m, f"frame {t}: exported mask")
    cf.close()


def test_long_free_run_with_spawning_and_deactivation():
    """50 frames at 160x128 with four moving objects: object models are spawned, deactivated (too few surfels / lost) and spawned
    again.  Free running against the oracle: model list, ids, poses, counts, surfel buffers and label masks identical every frame."""
    from co_fusion_amd import facade
    w, h = 160, 128  # the segmentation works on 16x16 superpixels: sizes are multiples of 16
    cam = synth.Camera.scaled(w, h)
    sc = synth.Scene(n_obj=4)
    ref = om.MultiPipeline(cam, conf_global=0.5, spawn_offset=2)
    cf = facade.CoFusion(w, h, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 17, conf_global_init=0.5, model_spawn_offset=2,
                         enable_multiple_models=1)
    counts = []
    for t in range(50):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        ref.process_frame(d, synth.rgb_to_rgba(rgb))
        cf.process_frame(d, rgb, timestamp=t)
        assert cf.num_models == len(ref.models), f"frame {t}: model count {cf.num_models} vs {len(ref.models)}"
        counts.append(len(ref.models))
        if t > 0:
            _same(cf.mask(), ref.mask, f"frame {t}: label mask")
        for i, m in enumerate(ref.models):
            info = cf.model_info(i)
            assert info["id"] == m.id, f"frame {t} model {i}: id"
            assert info["count"] == m.surfels.shape[0], f"frame {t} model {i}: count {info['count']} vs {m.surfels.shape[0]}"
            _same(info["pose"], m.pose, f"frame {t} model {i}: pose")
            if t % 5 == 4 or i > 0:
                _same(cf.model_download(i), m.surfels, f"frame {t} model {i}: surfels")
    cf.close()
    ups = sum(1 for a, b in zip(counts, counts[1:]) if b > a)
    downs = sum(1 for a, b in zip(counts, counts[1:]) if b < a)
    assert ups >= 2 and downs >= 2, f"spawn / deactivation not exercised: {counts}"


def test_device_frames_complete_overlap_changes_nothing():
    """device_frames_complete=1 lets the next frame's depth filter run beside the previous frame's fusion passes (double-buffered,
    auxiliary stream): poses, counts and surfel buffers must be those of the stream-ordered default, frame by frame."""
    import torch
    from co_fusion_amd import facade
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=2)
    dev = torch.device("cuda", 0)
    frames = []
    for t in range(10):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        frames.append((torch.from_numpy(d.astype(np.float32)).to(dev), torch.from_numpy(synth.rgb_to_rgba(rgb)).to(dev)))
    torch.cuda.synchronize()
    runs = []
    for flag in (0, 1):
        cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, max_surfels=1 << 19, conf_global_init=0.5, model_spawn_offset=2,
                             enable_multiple_models=1, device_frames_complete=flag)
        log = []
        for t, (d, c) in enumerate(frames):
            cf.process_frame_device(d, c, timestamp=t)
            log.append([(cf.model_info(i)["id"], cf.model_info(i)["count"], cf.model_info(i)["pose"].copy()) for i in range(cf.num_models)])
        final = [cf.model_download(i).copy() for i in range(cf.num_models)]
        cf.close()
        runs.append((log, final))
    (la, fa), (lb, fb) = runs
    assert len(fa) == len(fb) and len(fa) >= 2, "object models should have been spawned"
    for t, (a, b) in enumerate(zip(la, lb)):
        assert len(a) == len(b), f"frame {t}: model count"
        for (ia, ca, pa), (ib, cb, pb) in zip(a, b):
            assert ia == ib and ca == cb, f"frame {t}: id / count"
            _same(pa, pb, f"frame {t}: pose")
    for i, (a, b) in enumerate(zip(fa, fb)):
        _same(a, b, f"model {i}: final surfels")
