"""GPU parity of the surfel half of the hot path (bilateral, bootstrap, index map, splat prediction,
fill-in, fuse, clean) and of the whole `-static` frame loop: HIP through the C-ABI vs the CPU oracle.

Everything is compared bit-for-bit: surfel buffers, counts, index maps, prediction images, poses."""
import warnings

import numpy as np
import pytest

import common
import orc
import orc_pipeline as op
from co_fusion_amd import synth

pytestmark = pytest.mark.gpu
warnings.filterwarnings("ignore", category=RuntimeWarning)

W, H = 320, 240


@pytest.fixture(scope="module")
def ctx():
    from co_fusion_amd import api
    cam = synth.Camera.scaled(W, H)
    c = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy)
    yield c
    c.close()


def _same(a, b, what):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ok = a.view(np.uint8) == b.view(np.uint8)
    if not ok.all():
        # NaN payloads may differ; fall back to value equality with NaN == NaN
        if a.dtype.kind == "f":
            ok2 = (a == b) | ((a != a) & (b != b))
            assert ok2.all(), f"{what}: {np.count_nonzero(~ok2)} of {ok2.size} values differ"
        else:
            assert False, f"{what}: {np.count_nonzero(~ok)} bytes differ"


def test_bilateral_exact(ctx):
    from co_fusion_amd import model as M
    sc = synth.Scene(n_obj=2)
    cam = synth.Camera.scaled(W, H)
    d, _, _, _ = sc.render(cam, 3, noise=True)
    d[10:20, 30:60] = 0.0   # holes
    d[100, 100] = 7.0       # beyond the cutoff
    out = M.bilateral(ctx, ctx.to_device(d), 5.0).cpu().numpy()
    _same(out, op.bilateral(d, 5.0), "bilateral")
    # the two-pixels-per-lane kernel (even widths) against the one-pixel kernel's cases: holes and non-finite values at the image
    # borders (a tap outside the image must add exact zeros, a non-finite tap inside must poison its neighbours exactly as the
    # shader arithmetic does), a window wider than the image, an odd width (one-pixel kernel)
    e = d.copy()
    e[0, 5] = np.inf; e[H - 1, W - 3] = np.nan; e[50, 0] = np.inf; e[60, W - 1] = np.nan; e[:, 0:2] = 0.0; e[120:130, W - 7:] = 0.0
    _same(M.bilateral(ctx, ctx.to_device(e), 5.0).cpu().numpy(), op.bilateral(e, 5.0), "bilateral, borders and non-finite values")
    for shape in ((9, 16), (33, 18), (40, 161), (5, 320)):
        f = np.ascontiguousarray(d[:shape[0], :shape[1]])
        _same(M.bilateral(ctx, ctx.to_device(f), 5.0).cpu().numpy(), op.bilateral(f, 5.0), f"bilateral {shape}")



def test_index_map_exact(ctx):
    from co_fusion_amd import model as M
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=1)
    d, rgb, _, _ = sc.render(cam, 0, noise=True)
    rgba = synth.rgb_to_rgba(rgb)
    ocam = orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy)
    df = op.bilateral(d, 5.0)
    raw, n_raw = op.vertex_feedback(rgba, d, ocam, 1, 20.0)
    filt, _ = op.vertex_feedback(rgba, df, ocam, 1, 20.0)
    surf = op.model_initialise(raw, n_raw, filt)
    m = M.Model(ctx, 1 << 18)
    m.initialise(ctx.to_device(rgba), ctx.to_device(d), ctx.to_device(df), 1, 20.0)
    assert m.count() == surf.shape[0]
    _same(m.download_map(), surf, "initialise")
    pose = common.perturbed_pose(4, 0.01, 1.0)
    idx, vc, ct, nr = op.predict_indices(surf, pose, ocam, W, H, 20.0, 2, M.TIME_DELTA)
    m.predict_indices(pose, 2, 20.0)
    _same(m.buffer(0), idx, "index"); _same(m.buffer(1), vc, "vertConf"); _same(m.buffer(2), ct, "colorTime"); _same(m.buffer(3), nr, "normRad")
    assert (idx > 0).mean() > 0.5
    m.close()
