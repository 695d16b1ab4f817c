"""Python mirror of the surfel Model (Core/Model/Model.h:117-235) over the C-ABI (per-call access for tests and tools;
the frame loop itself is the C++ facade, co_fusion_amd/host/CoFusion.cpp)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .api import Context, Odometry, _f, _p

SURFEL = 12
TIME_DELTA = 2 ** 31 // 2 - 1  # openLoop: INT_MAX / 2 (GUI/MainController.cpp:328)

_BUF = {0: (np.uint32, 1), 1: (np.float32, 4), 2: (np.float32, 4), 3: (np.float32, 4), 4: (np.uint8, 4), 5: (np.float32, 4),
        6: (np.float32, 4), 7: (np.uint16, 1), 8: (np.float32, 4), 9: (np.float32, 4), 10: (np.uint8, 4)}


def bilateral(ctx: Context, depth, max_d):
    rows, cols = depth.shape
    out = ctx.empty((rows, cols))
    ctx._check(ctx.lib.cf_bilateral(ctx.h, _p(depth), cols, rows, C.c_float(max_d), _p(out)))
    return out


def fusion_weight(ctx: Context, pose, last_pose, mult):
    ctx.lib.cf_fusion_weight.restype = C.c_float
    return float(ctx.lib.cf_fusion_weight(_f(np.asarray(pose, np.float32).reshape(16)), _f(np.asarray(last_pose, np.float32).reshape(16)),
                                          C.c_float(mult)))


class Model:
    def __init__(self, ctx: Context, max_surfels=1 << 20):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._check(ctx.lib.cf_model_create(ctx.h, int(max_surfels), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.cf_model_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pose(self, pose):
        return _f(np.asarray(pose, np.float32).reshape(16))

    def initialise(self, rgba, depth_raw, depth_filt, time, max_depth):
        self.ctx._check(self.ctx.lib.cf_model_initialise(self.h, _p(rgba), _p(depth_raw), _p(depth_filt), time, C.c_float(max_depth)))

    def count(self):
        c = C.c_uint32()
        self.ctx._check(self.ctx.lib.cf_model_count(self.h, C.byref(c)))
        return c.value

    def predict_indices(self, pose, time, max_depth, time_delta=TIME_DELTA):
        self.ctx._check(self.ctx.lib.cf_model_predict_indices(self.h, self._pose(pose), time, C.c_float(max_depth), time_delta))

    def index_keys(self, pose, time, max_depth, surfel_begin, surfel_end, time_delta=TIME_DELTA):
        """first half of predict_indices for the surfel range [begin, end): int64 view of the u64 z-keys, [H, W] on the device"""
        keys = torch.empty((self.ctx.height, self.ctx.width), dtype=torch.int64, device=self.ctx.device)
        self.ctx._check(self.ctx.lib.cf_model_index_keys(self.h, self._pose(pose), time, C.c_float(max_depth), time_delta,
                                                         int(surfel_begin), int(surfel_end), C.c_void_p(keys.data_ptr())))
        return keys

    def index_resolve(self, pose, keys):
        """second half: resolve a (reduced) key map into the index-map textures"""
        self.ctx._check(self.ctx.lib.cf_model_index_resolve(self.h, self._pose(pose), C.c_void_p(keys.data_ptr())))

    def combined_predict(self, pose, max_depth, conf_threshold, time, max_time, time_delta=TIME_DELTA):
        self.ctx._check(self.ctx.lib.cf_model_combined_predict(self.h, self._pose(pose), C.c_float(max_depth), C.c_float(conf_threshold),
                                                               time, max_time, time_delta))

    def perform_fill_in(self, rgba, depth_filt, pass_geom=False, pass_rgb=False):
        self.ctx._check(self.ctx.lib.cf_model_perform_fill_in(self.h, _p(rgba), _p(depth_filt), int(pass_geom), int(pass_rgb)))

    def requires_fill_in(self, ratio=0.75):
        out = C.c_int()
        self.ctx._check(self.ctx.lib.cf_model_requires_fill_in(self.h, C.c_float(ratio), C.byref(out)))
        return bool(out.value)

    def fuse(self, pose, time, rgba, mask, depth_raw, depth_filt, max_depth, weighting, mask_id):
        self.ctx._check(self.ctx.lib.cf_model_fuse(self.h, self._pose(pose), time, _p(rgba), _p(mask), _p(depth_raw), _p(depth_filt),
                                                   C.c_float(max_depth), C.c_float(weighting), mask_id))

    def clean(self, pose, time, conf_threshold, outlier_coeff, depth_filt, mask, mask_id, time_delta=TIME_DELTA):
        c = C.c_uint32()
        self.ctx._check(self.ctx.lib.cf_model_clean(self.h, self._pose(pose), time, C.c_float(conf_threshold), C.c_float(outlier_coeff),
                                                    time_delta, _p(depth_filt), _p(mask), mask_id, C.byref(c)))
        return c.value

    def download_map(self):
        n = self.count()
        out = np.zeros((max(n, 1), SURFEL), np.float32)
        c = C.c_uint32()
        self.ctx._check(self.ctx.lib.cf_model_download_map(self.h, out.ctypes.data_as(C.c_void_p), n, C.byref(c)))
        return out[:n]

    def upload_map(self, surfels):
        s = np.ascontiguousarray(surfels, np.float32).reshape(-1, SURFEL)
        self.ctx._check(self.ctx.lib.cf_model_upload_map(self.h, s.ctypes.data_as(C.c_void_p), s.shape[0]))

    def tensor(self, which):
        """Zero-copy torch view of a projection buffer (stays valid while the model lives)."""
        ptr = C.c_void_p(); nbytes = C.c_uint64()
        self.ctx._check(self.ctx.lib.cf_model_buffer(self.h, which, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def buffer(self, which):
        ptr, nbytes = self.tensor(which)
        host = np.empty(nbytes, np.uint8)
        self.ctx._check(self.ctx.lib.cf_memcpy_d2h(self.ctx.h, host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_uint64(nbytes)))
        h, w = self.ctx.height, self.ctx.width
        if which == 11:
            return host.view(np.float32).reshape(-1, SURFEL)
        dt, ch = _BUF[which]
        a = host.view(dt)
        return a.reshape(h, w, ch) if ch > 1 else a.reshape(h, w)


class _DevView:
    """Minimal stand-in that lets _p() pass a raw device pointer owned by the library."""

    def __init__(self, ptr):
        self._ptr = ptr
        self.is_cuda = True

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self._ptr
