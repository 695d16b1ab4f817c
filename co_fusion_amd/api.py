"""Python mirror of the reference interface for the hot path, over the C-ABI.

PyTorch is used for device memory and streams only (tensors are passed to the library as raw
device pointers); every computation happens in the hand-written HIP kernels.

Names follow the reference: `Odometry` mirrors RGBDOdometry (Core/Utils/RGBDOdometry.h:42-60),
the free functions mirror Core/Cuda/cudafuncs.cuh:64-193.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as _libmod

NUM_PYRS = 3


class Cam(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]

    def level(self, l):
        d = float(1 << l)
        return Cam(self.fx / d, self.fy / d, self.cx / d, self.cy / d)


class Config(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("device", C.c_int), ("max_models", C.c_int), ("max_surfels", C.c_int)]


class TrackOpts(C.Structure):
    _fields_ = [("rgb_only", C.c_int), ("pyramid", C.c_int), ("fast_odom", C.c_int), ("so3", C.c_int),
                ("icp_weight", C.c_float)]


class TrackStats(C.Structure):
    _fields_ = [("last_icp_error", C.c_float), ("last_icp_count", C.c_float), ("last_rgb_error", C.c_float),
                ("last_rgb_count", C.c_float), ("last_so3_error", C.c_float), ("last_so3_count", C.c_float),
                ("lastA", C.c_double * 36), ("lastb", C.c_double * 6), ("so3_iterations", C.c_int), ("fault", C.c_int), ("cull_box", C.c_int * 4)]


class Profile(C.Structure):
    _fields_ = [("icp_ms_total", C.c_double), ("icp_launches", C.c_uint64), ("icp_bytes", C.c_uint64),
                ("surfel_ms_total", C.c_double), ("surfel_calls", C.c_uint64), ("surfel_bytes", C.c_uint64)]


DATATERM = np.dtype([("zero_x", "<i2"), ("zero_y", "<i2"), ("one_x", "<i2"), ("one_y", "<i2"), ("diff", "<f4"),
                     ("valid", "<i4")])


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def _f(a):
    return (C.c_float * len(a))(*[float(x) for x in a])


class CofusionError(RuntimeError):
    pass


class Context:
    """One context per GPU (cf_ctx): owns scratch memory; work is enqueued on torch's current stream."""

    def __init__(self, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, device=0, max_models=8,
                 max_surfels=3072 * 3072):
        if not torch.cuda.is_available():
            raise CofusionError("no GPU visible: the Co-Fusion hot path has no CPU fallback")
        self.lib = _libmod.load()
        self.width, self.height = width, height
        self.cam = Cam(fx, fy, cx, cy)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        cfg = Config(width, height, fx, fy, cx, cy, device, max_models, max_surfels)
        h = C.c_void_p()
        rc = self.lib.cf_create(C.byref(cfg), C.byref(h))
        self.h = h
        self._check(rc)
        self._check(self.lib.cf_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.cf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.cf_last_error(self.h) if self.h else b""
            raise CofusionError(f"cofusion_hip error {rc}: {msg.decode() if msg else ''}")

    def synchronize(self):
        self._check(self.lib.cf_synchronize(self.h))

    # ---- the library's own RCCL communicator (csrc/rccl_comm.hip) ----
    @staticmethod
    def rccl_unique_id():
        buf = (C.c_ubyte * 128)()
        if _libmod.load().cf_rccl_unique_id(buf) != 0:
            raise CofusionError("cf_rccl_unique_id failed")
        return bytes(buf)

    def rccl_init(self, unique_id, rank, world):
        self._check(self.lib.cf_rccl_init(self.h, C.c_char_p(unique_id), int(rank), int(world)))

    def rccl_allreduce(self, tensor, op=0):
        """in place on the context's stream: op 0 = SUM of int64 words, op 1 = MIN of unsigned 64-bit words"""
        assert tensor.element_size() == 8 and tensor.is_contiguous()
        self._check(self.lib.cf_rccl_allreduce(self.h, C.c_void_p(tensor.data_ptr()), C.c_uint64(tensor.numel()), int(op), None))

    def rccl_broadcast(self, tensor, root=0):
        self._check(self.lib.cf_rccl_broadcast(self.h, C.c_void_p(tensor.data_ptr()), C.c_uint64(tensor.numel() * tensor.element_size()), int(root), None))

    def rccl_info(self):
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        rc = self.lib.cf_rccl_info(self.h, C.byref(r), C.byref(w), C.byref(v))
        return dict(active=rc == 0, rank=r.value, world=w.value, version=v.value)

    def empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    # ---- map preparation (cudafuncs.cuh) -------------------------------------------------
    def create_vmap(self, depth, cam, cutoff):
        rows, cols = depth.shape
        out = torch.zeros((3 * rows, cols), dtype=torch.float32, device=self.device)
        self._check(self.lib.cf_create_vmap(self.h, _p(depth), cols, rows, cam, C.c_float(cutoff), _p(out)))
        return out

    def create_nmap(self, vmap):
        rows, cols = vmap.shape[0] // 3, vmap.shape[1]
        out = torch.zeros_like(vmap)
        self._check(self.lib.cf_create_nmap(self.h, _p(vmap), cols, rows, _p(out)))
        return out

    def copy_maps(self, v4, n4):
        rows, cols = v4.shape[:2]
        v = self.empty((3 * rows, cols))
        n = self.empty((3 * rows, cols))
        self._check(self.lib.cf_copy_maps(self.h, _p(v4), _p(n4), cols, rows, _p(v), _p(n)))
        return v, n

    def resize_map(self, m, normalize):
        rows, cols = m.shape[0] // 3, m.shape[1]
        out = torch.zeros((3 * (rows // 2), cols // 2), dtype=torch.float32, device=self.device)
        self._check(self.lib.cf_resize_map(self.h, _p(m), cols, rows, _p(out), int(normalize)))
        return out

    def transform_maps(self, v, n, R, t):
        rows, cols = v.shape[0] // 3, v.shape[1]
        self._check(self.lib.cf_transform_maps(self.h, _p(v), _p(n), cols, rows, _f(np.asarray(R).reshape(9)),
                                               _f(np.asarray(t).reshape(3))))

    def vertices_to_depth(self, v4, cutoff):
        rows, cols = v4.shape[:2]
        out = self.empty((rows, cols))
        self._check(self.lib.cf_vertices_to_depth(self.h, _p(v4), cols, rows, C.c_float(cutoff), _p(out)))
        return out

    def pyrdown_gauss_f32(self, src):
        rows, cols = src.shape
        out = self.empty((rows // 2, cols // 2))
        self._check(self.lib.cf_pyrdown_gauss_f32(self.h, _p(src), cols, rows, _p(out)))
        return out

    def pyrdown_gauss_u8(self, src):
        rows, cols = src.shape
        out = self.empty((rows // 2, cols // 2), torch.uint8)
        self._check(self.lib.cf_pyrdown_gauss_u8(self.h, _p(src), cols, rows, _p(out)))
        return out

    def rgba_to_intensity(self, rgba):
        rows, cols = rgba.shape[:2]
        out = self.empty((rows, cols), torch.uint8)
        self._check(self.lib.cf_rgba_to_intensity(self.h, _p(rgba), cols, rows, _p(out)))
        return out

    def sobel(self, img):
        rows, cols = img.shape
        dx = self.empty((rows, cols), torch.int16)
        dy = self.empty((rows, cols), torch.int16)
        self._check(self.lib.cf_sobel(self.h, _p(img), cols, rows, _p(dx), _p(dy)))
        return dx, dy

    def project_cloud(self, depth, cam_level):
        rows, cols = depth.shape
        out = self.empty((rows, cols, 3))
        self._check(self.lib.cf_project_cloud(self.h, _p(depth), cols, rows, cam_level, _p(out)))
        return out

    def depth_pyramid(self, depth):
        rows, cols = depth.shape
        l1 = self.empty((rows // 2, cols // 2))
        l2 = self.empty((rows // 4, cols // 4))
        self._check(self.lib.cf_depth_pyramid(self.h, _p(depth), cols, rows, _p(l1), _p(l2)))
        return [depth, l1, l2]

    # ---- reductions (cudafuncs.cuh) ---------------------------------------------------------
    def icp_step(self, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, dist_thres,
                 angle_thres, err_surface=None):
        rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
        A = (C.c_float * 36)(); b = (C.c_float * 6)(); res = (C.c_float * 2)(); sums = (C.c_int64 * 32)()
        self._check(self.lib.cf_icp_step(self.h, _f(np.asarray(Rcurr).reshape(9)), _f(tcurr), _p(vmap_curr),
                                         _p(nmap_curr), _f(np.asarray(Rprev_inv).reshape(9)), _f(tprev), cam,
                                         _p(vmap_g_prev), _p(nmap_g_prev), C.c_float(dist_thres), C.c_float(angle_thres),
                                         cols, rows, A, b, res, sums, _p(err_surface)))
        return (np.array(A, np.float32).reshape(6, 6), np.array(b, np.float32), np.array(res, np.float32),
                np.array(sums, np.int64))

    def icp_step_band(self, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, dist_thres,
                      angle_thres, row_begin, row_end):
        """icpStep over the row band [row_begin, row_end) only: returns the exact int64[32] sums of the band"""
        rows, cols = vmap_curr.shape[0] // 3, vmap_curr.shape[1]
        A = (C.c_float * 36)(); b = (C.c_float * 6)(); res = (C.c_float * 2)(); sums = (C.c_int64 * 32)()
        self._check(self.lib.cf_icp_step_band(self.h, _f(np.asarray(Rcurr).reshape(9)), _f(tcurr), _p(vmap_curr),
                                              _p(nmap_curr), _f(np.asarray(Rprev_inv).reshape(9)), _f(tprev), cam,
                                              _p(vmap_g_prev), _p(nmap_g_prev), C.c_float(dist_thres), C.c_float(angle_thres),
                                              cols, rows, int(row_begin), int(row_end), A, b, res, sums, None))
        return np.array(sums, np.int64)

    def rgb_residual(self, min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_depth_delta, kt,
                     krkinv):
        rows, cols = next_image.shape
        corres = torch.zeros((rows * cols, 16), dtype=torch.uint8, device=self.device)
        sig = C.c_int(); cnt = C.c_int()
        self._check(self.lib.cf_rgb_residual(self.h, C.c_float(min_scale), _p(dIdx), _p(dIdy), _p(last_depth),
                                             _p(next_depth), _p(last_image), _p(next_image), _p(corres),
                                             C.c_float(max_depth_delta), _f(kt), _f(np.asarray(krkinv).reshape(9)),
                                             cols, rows, C.byref(sig), C.byref(cnt)))
        return corres, sig.value, cnt.value

    def rgb_step(self, corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale):
        rows, cols = dIdx.shape
        A = (C.c_float * 36)(); b = (C.c_float * 6)(); sums = (C.c_int64 * 32)()
        self._check(self.lib.cf_rgb_step(self.h, _p(corres), C.c_float(sigma), _p(cloud), C.c_float(fx), C.c_float(fy),
                                         _p(dIdx), _p(dIdy), C.c_float(sobel_scale), cols, rows, A, b, sums))
        return np.array(A, np.float32).reshape(6, 6), np.array(b, np.float32), np.array(sums, np.int64)

    def so3_step(self, last_image, next_image, basis, kinv, krlr):
        rows, cols = next_image.shape
        A = (C.c_float * 9)(); b = (C.c_float * 3)(); res = (C.c_float * 2)(); sums = (C.c_int64 * 16)()
        self._check(self.lib.cf_so3_step(self.h, _p(last_image), _p(next_image), _f(np.asarray(basis).reshape(9)),
                                         _f(np.asarray(kinv).reshape(9)), _f(np.asarray(krlr).reshape(9)), cols, rows,
                                         A, b, res, sums))
        return (np.array(A, np.float32).reshape(3, 3), np.array(b, np.float32), np.array(res, np.float32),
                np.array(sums, np.int64))

    def set_icp_launch(self, threads, ppt):
        self._check(self.lib.cf_set_icp_launch(self.h, threads, ppt))

    def set_icp_arith(self, mode):
        """rounding specification of the ICP sums: 0 / "product" (default), 1 / "gram" or 2 / "reference" (the reference's own f32 trees and host loop; include/cofusion_hip.h: cf_set_icp_arith)"""
        self._check(self.lib.cf_set_icp_arith(self.h, {"product": 0, "gram": 1, "reference": 2}.get(mode, mode)))

    def profile_enable(self, on=True):
        self._check(self.lib.cf_profile_enable(self.h, int(on)))

    def profile_read(self, reset=True):
        p = Profile()
        self._check(self.lib.cf_profile_read(self.h, C.byref(p), int(reset)))
        return p


class Odometry:
    """Device-resident RGBDOdometry (Core/Utils/RGBDOdometry.h:42-60)."""

    _BUF = {0: (np.float32, 3), 1: (np.float32, 3), 2: (np.float32, 3), 3: (np.float32, 3), 4: (np.float32, 1),
            5: (np.float32, 1), 6: (np.uint8, 1), 7: (np.uint8, 1), 8: (np.uint8, 1), 9: (np.int16, 1),
            10: (np.int16, 1)}

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.h = C.c_void_p()
        ctx._check(ctx.lib.cf_odom_create(ctx.h, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.lib.cf_odom_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init_icp_model(self, pred_v4, pred_n4, pose):
        self.ctx._check(self.ctx.lib.cf_odom_init_icp_model(self.h, _p(pred_v4), _p(pred_n4),
                                                            _f(np.asarray(pose, np.float32).reshape(16))))

    def init_rgb_model(self, rgba):
        self.ctx._check(self.ctx.lib.cf_odom_init_rgb_model(self.h, _p(rgba)))

    def set_band(self, row_begin, row_end, add_counts=1):
        """this rank's rows of the model's reductions (cf_odom_set_band); (0, 0): all rows, no collective"""
        self.ctx._check(self.ctx.lib.cf_odom_set_band(self.h, int(row_begin), int(row_end), int(add_counts)))

    def set_culling(self, on=True):
        self.ctx._check(self.ctx.lib.cf_odom_set_culling(self.h, int(bool(on))))

    def init_rgb(self, rgba):
        self.ctx._check(self.ctx.lib.cf_odom_init_rgb(self.h, _p(rgba)))

    def init_first_rgb(self, rgba):
        self.ctx._check(self.ctx.lib.cf_odom_init_first_rgb(self.h, _p(rgba)))

    def init_icp(self, depth_pyr, cutoff):
        arr = (C.c_void_p * 3)(*[d.data_ptr() for d in depth_pyr])
        self._keep = depth_pyr
        self.ctx._check(self.ctx.lib.cf_odom_init_icp(self.h, arr, C.c_float(cutoff)))

    def track(self, trans, rot, rgb_only=False, icp_weight=10.0, pyramid=True, fast_odom=False, so3=True,
              err_surface=None):
        t = _f(np.asarray(trans, np.float32).reshape(3))
        r = _f(np.asarray(rot, np.float32).reshape(9))
        opts = TrackOpts(int(rgb_only), int(pyramid), int(fast_odom), int(so3), icp_weight)
        st = TrackStats()
        self.ctx._check(self.ctx.lib.cf_odom_get_incremental_transformation(self.h, t, r, C.byref(opts),
                                                                            _p(err_surface), C.byref(st)))
        return np.array(t, np.float32), np.array(r, np.float32).reshape(3, 3), st

    def bench_icp(self, level, iters=200):
        us = C.c_float()
        self.ctx._check(self.ctx.lib.cf_odom_bench_icp(self.h, level, iters, C.byref(us)))
        return us.value

    def buffer(self, which, level):
        ptr = C.c_void_p(); nbytes = C.c_uint64()
        self.ctx._check(self.ctx.lib.cf_odom_buffer(self.h, which, level, C.byref(ptr), C.byref(nbytes)))
        host = np.empty(nbytes.value, np.uint8)
        self.ctx._check(self.ctx.lib.cf_memcpy_d2h(self.ctx.h, host.ctypes.data_as(C.c_void_p), ptr, nbytes))
        w, h = self.ctx.width >> level, self.ctx.height >> level
        if which == 11:
            return host.view(np.float32).reshape(h, w, 3)
        if which == 12:
            return host.view(DATATERM).reshape(h * w)
        dt, planes = self._BUF[which]
        return host.view(dt).reshape(planes * h, w)
