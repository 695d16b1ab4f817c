"""ctypes binding of the CPU oracle (oracle/_build/liborc.so).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# ORC_LIB: another build of the same sources (bench.py times oracle/_build/liborc_native.so, -O3 -march=native)
_LIB = os.environ.get("ORC_LIB") or os.path.join(_ROOT, "oracle", "_build", "liborc.so")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def _load():
    if not os.path.exists(_LIB):
        build()
    return C.CDLL(_LIB)


lib = _load()


def usable_cpus() -> int:
    """cores this process may really use: affinity mask AND the cgroup quota (a container can show 256 CPUs and grant 8)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def set_threads(n: int) -> None:
    C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


# the oracle's OpenMP loops (bilateral filter, full-size tracking passes): never more threads than cores really granted, and a small
# team by default -- the tests run many short calls
set_threads(min(8, usable_cpus()))


class Cam(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]

    def level(self, l):
        d = float(1 << l)
        return Cam(self.fx / d, self.fy / d, self.cx / d, self.cy / d)


class TrackOpts(C.Structure):
    _fields_ = [("rgb_only", C.c_int), ("pyramid", C.c_int), ("fast_odom", C.c_int), ("so3", C.c_int),
                ("icp_weight", C.c_float)]


class TrackStats(C.Structure):
    _fields_ = [("last_icp_error", C.c_float), ("last_icp_count", C.c_float), ("last_rgb_error", C.c_float),
                ("last_rgb_count", C.c_float), ("last_so3_error", C.c_float), ("last_so3_count", C.c_float),
                ("lastA", C.c_double * 36), ("lastb", C.c_double * 6), ("so3_iterations", C.c_int)]


DATATERM = np.dtype([("zero_x", "<i2"), ("zero_y", "<i2"), ("one_x", "<i2"), ("one_y", "<i2"), ("diff", "<f4"),
                     ("valid", "<i4")])


def P(a):
    """pointer to a contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


lib.orc_odom_create.restype = C.c_void_p
lib.orc_odom_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
lib.orc_odom_destroy.argtypes = [C.c_void_p]
lib.orc_odom_buffer.restype = C.c_void_p
lib.orc_odom_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int]


class Odometry:
    """Mirror of RGBDOdometry (Core/Utils/RGBDOdometry.h:42-60) on the oracle."""

    def __init__(self, w, h, cx, cy, fx, fy):
        self.w, self.h = w, h
        self.h_ = lib.orc_odom_create(w, h, cx, cy, fx, fy)

    def __del__(self):
        if getattr(self, "h_", None):
            lib.orc_odom_destroy(C.c_void_p(self.h_))
            self.h_ = None

    def init_icp_model(self, v4, n4, pose):
        pose = f32(pose).reshape(16)
        lib.orc_odom_init_icp_model(C.c_void_p(self.h_), P(f32(v4)), P(f32(n4)), P(pose))

    def init_rgb_model(self, rgba):
        lib.orc_odom_init_rgb_model(C.c_void_p(self.h_), P(u8(rgba)))

    def init_icp(self, depth_pyr, cutoff):
        keep = [f32(d) for d in depth_pyr]
        arr = (C.c_void_p * 3)(*[d.ctypes.data for d in keep])
        lib.orc_odom_init_icp(C.c_void_p(self.h_), arr, C.c_float(cutoff))

    def init_rgb(self, rgba):
        lib.orc_odom_init_rgb(C.c_void_p(self.h_), P(u8(rgba)))

    def init_first_rgb(self, rgba):
        lib.orc_odom_init_first_rgb(C.c_void_p(self.h_), P(u8(rgba)))

    def track(self, trans, rot, rgb_only=False, icp_weight=10.0, pyramid=True, fast_odom=False, so3=True,
              err_surface=None):
        trans = f32(trans).copy().reshape(3)
        rot = f32(rot).copy().reshape(9)
        opts = TrackOpts(int(rgb_only), int(pyramid), int(fast_odom), int(so3), icp_weight)
        st = TrackStats()
        lib.orc_odom_get_incremental_transformation(C.c_void_p(self.h_), P(trans), P(rot), C.byref(opts),
                                                    P(err_surface), C.byref(st))
        return trans, rot.reshape(3, 3), st

    def buffer(self, which, level):
        w, h = self.w >> level, self.h >> level
        ptr = lib.orc_odom_buffer(C.c_void_p(self.h_), which, level)
        if which <= 3:
            shape, dt = (3 * h, w), np.float32
        elif which <= 5:
            shape, dt = (h, w), np.float32
        elif which <= 8:
            shape, dt = (h, w), np.uint8
        elif which <= 10:
            shape, dt = (h, w), np.int16
        elif which == 11:
            shape, dt = (h, w, 3), np.float32
        else:
            shape, dt = (h * w,), DATATERM
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        buf = (C.c_char * n).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).reshape(shape).copy()


def depth_pyramid(depth):
    h, w = depth.shape
    l1 = np.empty((h // 2, w // 2), np.float32)
    l2 = np.empty((h // 4, w // 4), np.float32)
    lib.orc_depth_pyramid(P(f32(depth)), w, h, P(l1), P(l2))
    return [f32(depth), l1, l2]


# ---------------------------------------------------------------- stand-alone functions ----
def create_vmap(depth, cam, cutoff):
    h, w = depth.shape
    out = np.zeros((3 * h, w), np.float32)
    lib.orc_create_vmap(P(f32(depth)), w, h, cam, C.c_float(cutoff), P(out))
    return out


def create_nmap(vmap):
    h, w = vmap.shape[0] // 3, vmap.shape[1]
    out = np.zeros_like(vmap)
    lib.orc_create_nmap(P(f32(vmap)), w, h, P(out))
    return out


def copy_maps(v4, n4):
    h, w = v4.shape[:2]
    v = np.empty((3 * h, w), np.float32); n = np.empty((3 * h, w), np.float32)
    lib.orc_copy_maps(P(f32(v4)), P(f32(n4)), w, h, P(v), P(n))
    return v, n


def resize_map(m, normalize):
    h, w = m.shape[0] // 3, m.shape[1]
    out = np.zeros((3 * (h // 2), w // 2), np.float32)
    lib.orc_resize_map(P(f32(m)), w, h, P(out), int(normalize))
    return out


def transform_maps(v, n, R, t):
    v = f32(v).copy(); n = f32(n).copy()
    h, w = v.shape[0] // 3, v.shape[1]
    lib.orc_transform_maps(P(v), P(n), w, h, P(f32(R).reshape(9)), P(f32(t).reshape(3)))
    return v, n


def vertices_to_depth(v4, cutoff):
    h, w = v4.shape[:2]
    out = np.empty((h, w), np.float32)
    lib.orc_vertices_to_depth(P(f32(v4)), w, h, C.c_float(cutoff), P(out))
    return out


def pyrdown_gauss_f32(src):
    h, w = src.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib.orc_pyrdown_gauss_f32(P(f32(src)), w, h, P(out))
    return out


def pyrdown_gauss_u8(src):
    h, w = src.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib.orc_pyrdown_gauss_u8(P(u8(src)), w, h, P(out))
    return out


def rgba_to_intensity(rgba):
    h, w = rgba.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib.orc_rgba_to_intensity(P(u8(rgba)), w, h, P(out))
    return out


def sobel(img):
    h, w = img.shape
    dx = np.empty((h, w), np.int16); dy = np.empty((h, w), np.int16)
    lib.orc_sobel(P(u8(img)), w, h, P(dx), P(dy))
    return dx, dy


def project_cloud(depth, cam_level):
    h, w = depth.shape
    out = np.empty((h, w, 3), np.float32)
    lib.orc_project_cloud(P(f32(depth)), w, h, cam_level, P(out))
    return out


def rgb_fix_bits(sigma):
    """fraction bits of the RGB step's fixed-point sums for this sigma (oracle/orc_math.h orc_rgb_fix_bits)."""
    return int(lib.orc_rgb_fix_bits_of(C.c_float(sigma)))


def se3_to_host(sums, F=32):
    A = np.zeros(36, np.float32); b = np.zeros(6, np.float32); r = np.zeros(2, np.float32)
    lib.orc_se3_sums_to_host(P(np.ascontiguousarray(sums, np.int64)), F, P(A), P(b), P(r))
    return A.reshape(6, 6), b, r


def set_icp_arith(mode):
    """rounding specification of the ICP sums, process-global (oracle/orc.h): 0 / "product" (default), 1 / "gram", or 2 / "reference" (the reference's own
    f32 trees and host algebra order: every reduction of the tracker, not only the ICP sums)"""
    lib.orc_set_icp_arith({"product": 0, "gram": 1, "reference": 2}.get(mode, mode))


def get_icp_arith():
    return int(lib.orc_get_icp_arith())


def icp_sums_to_host(sums):
    """the ICP sums under the current rounding specification -> A, b, residual"""
    A = np.zeros(36, np.float32); b = np.zeros(6, np.float32); r = np.zeros(2, np.float32)
    lib.orc_icp_sums_to_host(P(np.ascontiguousarray(sums, np.int64)), P(A), P(b), P(r))
    return A.reshape(6, 6), b, r


def so3_to_host(sums, F=12):
    A = np.zeros(9, np.float32); b = np.zeros(3, np.float32); r = np.zeros(2, np.float32)
    lib.orc_so3_sums_to_host(P(np.ascontiguousarray(sums, np.int64)), F, P(A), P(b), P(r))
    return A.reshape(3, 3), b, r


def icp_step(Rcurr, tcurr, vc, nc, Rprev_inv, tprev, cam, vp, np_, dist, angle, want_err=False):
    h, w = vc.shape[0] // 3, vc.shape[1]
    sums = np.zeros(32, np.int64)
    err = np.zeros((h, w), np.float32) if want_err else None
    lib.orc_icp_step(P(f32(Rcurr).reshape(9)), P(f32(tcurr)), P(f32(vc)), P(f32(nc)), P(f32(Rprev_inv).reshape(9)),
                     P(f32(tprev)), cam, P(f32(vp)), P(f32(np_)), C.c_float(dist), C.c_float(angle), w, h, P(sums), P(err))
    return sums, err


def icp_step_f32tree(Rcurr, tcurr, vc, nc, Rprev_inv, tprev, cam, vp, np_, dist, angle, threads, blocks):
    h, w = vc.shape[0] // 3, vc.shape[1]
    out = np.zeros(29, np.float32)
    lib.orc_icp_step_f32tree(P(f32(Rcurr).reshape(9)), P(f32(tcurr)), P(f32(vc)), P(f32(nc)),
                             P(f32(Rprev_inv).reshape(9)), P(f32(tprev)), cam, P(f32(vp)), P(f32(np_)), C.c_float(dist),
                             C.c_float(angle), w, h, threads, blocks, P(out))
    return out


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_dd, kt, krkinv):
    h, w = next_image.shape
    corres = np.zeros(h * w, DATATERM)
    sig = C.c_int(); cnt = C.c_int()
    lib.orc_rgb_residual(C.c_float(min_scale), P(np.ascontiguousarray(dIdx, np.int16)),
                         P(np.ascontiguousarray(dIdy, np.int16)), P(f32(last_depth)), P(f32(next_depth)),
                         P(u8(last_image)), P(u8(next_image)), P(corres), C.c_float(max_dd), P(f32(kt)),
                         P(f32(krkinv).reshape(9)), w, h, C.byref(sig), C.byref(cnt))
    return corres, sig.value, cnt.value


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale):
    h, w = dIdx.shape
    sums = np.zeros(32, np.int64)
    lib.orc_rgb_step(P(corres), C.c_float(sigma), P(f32(cloud)), C.c_float(fx), C.c_float(fy),
                     P(np.ascontiguousarray(dIdx, np.int16)), P(np.ascontiguousarray(dIdy, np.int16)),
                     C.c_float(sobel_scale), w, h, P(sums))
    return sums


def so3_step(last_image, next_image, basis, kinv, krlr):
    h, w = next_image.shape
    sums = np.zeros(16, np.int64)
    lib.orc_so3_step(P(u8(last_image)), P(u8(next_image)), P(f32(basis).reshape(9)), P(f32(kinv).reshape(9)),
                     P(f32(krlr).reshape(9)), w, h, P(sums))
    return sums


def _se3_from29(v):
    """JtJJtrSE3's 29 floats -> A, b, residual the way icpStep / rgbStep unpack them (reduce.cu:481-498)."""
    A = np.zeros((6, 6), np.float32); b = np.zeros(6, np.float32); k = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                b[i] = v[k]
            else:
                A[i, j] = A[j, i] = v[k]
            k += 1
    return A, b, np.array(v[27:29], np.float32)


def icp_step_ref_order(*a, threads=128, blocks=112):
    return _se3_from29(icp_step_f32tree(*a, threads, blocks))


def rgb_step_ref_order(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale, threads=128, blocks=112):
    h, w = dIdx.shape
    out = np.zeros(29, np.float32)
    lib.orc_rgb_step_f32tree(P(corres), C.c_float(sigma), P(f32(cloud)), C.c_float(fx), C.c_float(fy),
                             P(np.ascontiguousarray(dIdx, np.int16)), P(np.ascontiguousarray(dIdy, np.int16)),
                             C.c_float(sobel_scale), w, h, threads, blocks, P(out))
    return _se3_from29(out)[:2]


def so3_step_ref_order(last_image, next_image, basis, kinv, krlr, threads=160, blocks=64):
    h, w = next_image.shape
    out = np.zeros(11, np.float32)
    lib.orc_so3_step_f32tree(P(u8(last_image)), P(u8(next_image)), P(f32(basis).reshape(9)), P(f32(kinv).reshape(9)),
                             P(f32(krlr).reshape(9)), w, h, threads, blocks, P(out))
    A = np.zeros((3, 3), np.float32); b = np.zeros(3, np.float32); k = 0
    for i in range(3):
        for j in range(i, 4):
            if j == 3:
                b[i] = out[k]
            else:
                A[i, j] = A[j, i] = out[k]
            k += 1
    return A, b, out[9:11].copy()
