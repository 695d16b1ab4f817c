"""ctypes binding of oracle/_ref/libcofusion_ref.so: the reference's OWN CUDA sources (Core/Cuda/reduce.cu, cudafuncs.cu)
compiled by g++ under the CPU SIMT emulator of oracle/ref_shim (build: oracle/ref_shim/build_ref.py).
Test infrastructure only; same function names and array layouts as tests/orc.py so the pin tests read 1:1."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(_ROOT, "oracle", "_ref", "libcofusion_ref.so")

# GPUConfig.h:51-58 launch shapes of the reference's reductions (threads, blocks)
ICP_LAUNCH = (128, 112)
RGB_LAUNCH = (128, 112)
RES_LAUNCH = (256, 336)
SO3_LAUNCH = (160, 64)

DATATERM = np.dtype([("zero_x", "<i2"), ("zero_y", "<i2"), ("one_x", "<i2"), ("one_y", "<i2"), ("diff", "<f4"),
                     ("valid", "u1"), ("pad", "u1", 3)])


def available() -> bool:
    if not os.path.exists(LIB) and os.path.isdir("/root/reference/Core/Cuda"):
        subprocess.check_call([sys.executable, os.path.join(_ROOT, "oracle", "ref_shim", "build_ref.py")])
    return os.path.exists(LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        assert available(), "oracle/_ref/libcofusion_ref.so missing (needs /root/reference to build)"
        _lib = C.CDLL(LIB)
        assert _lib.ref_sizeof_dataterm() == DATATERM.itemsize
    return _lib


def P(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def i16(a):
    return np.ascontiguousarray(a, dtype=np.int16)


F = C.c_float


def create_vmap(depth, cam, cutoff):
    h, w = depth.shape
    out = np.zeros((3 * h, w), np.float32)
    lib().ref_create_vmap(P(f32(depth)), w, h, F(cam.fx), F(cam.fy), F(cam.cx), F(cam.cy), F(cutoff), P(out))
    return out


def create_nmap(vmap):
    h, w = vmap.shape[0] // 3, vmap.shape[1]
    out = np.zeros_like(vmap)
    lib().ref_create_nmap(P(f32(vmap)), w, h, P(out))
    return out


def copy_maps(v4, n4):
    h, w = v4.shape[:2]
    v = np.empty((3 * h, w), np.float32); n = np.empty((3 * h, w), np.float32)
    lib().ref_copy_maps(P(f32(v4)), P(f32(n4)), w, h, P(v), P(n))
    return v, n


def resize_map(m, normalize):
    h, w = m.shape[0] // 3, m.shape[1]
    out = np.zeros((3 * (h // 2), w // 2), np.float32)
    lib().ref_resize_map(P(f32(m)), w, h, int(normalize), P(out))
    return out


def transform_maps(v, n, R, t):
    v = f32(v).copy(); n = f32(n).copy()
    h, w = v.shape[0] // 3, v.shape[1]
    lib().ref_transform_maps(P(v), P(n), w, h, P(f32(R).reshape(9)), P(f32(t).reshape(3)))
    return v, n


def vertices_to_depth(v4, cutoff):
    h, w = v4.shape[:2]
    out = np.empty((h, w), np.float32)
    lib().ref_vertices_to_depth(P(f32(v4)), w, h, F(cutoff), P(out))
    return out


def pyrdown_gauss_f32(src):
    h, w = src.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().ref_pyrdown_gauss_f32(P(f32(src)), w, h, P(out))
    return out


def pyrdown_gauss_u8(src):
    h, w = src.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().ref_pyrdown_gauss_u8(P(u8(src)), w, h, P(out))
    return out


def rgba_to_intensity(rgba):
    h, w = rgba.shape[:2]
    out = np.empty((h, w), np.uint8)
    lib().ref_rgba_to_intensity(P(u8(rgba)), w, h, P(out))
    return out


def sobel(img):
    h, w = img.shape
    dx = np.empty((h, w), np.int16); dy = np.empty((h, w), np.int16)
    lib().ref_sobel(P(u8(img)), w, h, P(dx), P(dy))
    return dx, dy


def project_cloud(depth, cam0, level):
    h, w = depth.shape
    out = np.empty((h, w, 3), np.float32)
    lib().ref_project_cloud(P(f32(depth)), w, h, F(cam0.fx), F(cam0.fy), F(cam0.cx), F(cam0.cy), level, P(out))
    return out


def icp_step(Rcurr, tcurr, vc, nc, Rprev_inv, tprev, cam, vp, np_, dist, angle, want_err=False, launch=ICP_LAUNCH):
    h, w = vc.shape[0] // 3, vc.shape[1]
    A = np.zeros(36, np.float32); b = np.zeros(6, np.float32); r = np.zeros(2, np.float32)
    err = np.zeros((h, w), np.float32) if want_err else None
    lib().ref_icp_step(P(f32(Rcurr).reshape(9)), P(f32(tcurr)), P(f32(vc)), P(f32(nc)), P(f32(Rprev_inv).reshape(9)),
                       P(f32(tprev)), F(cam.fx), F(cam.fy), F(cam.cx), F(cam.cy), P(f32(vp)), P(f32(np_)), F(dist), F(angle),
                       w, h, launch[0], launch[1], P(A), P(b), P(r), P(err))
    return A.reshape(6, 6), b, r, err


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, max_dd, kt, krkinv, want_err=False,
                 launch=RES_LAUNCH):
    h, w = next_image.shape
    corres = np.zeros(h * w, DATATERM)
    sig = C.c_int(); cnt = C.c_int()
    err = np.zeros((h, w), np.float32) if want_err else None
    lib().ref_rgb_residual(F(min_scale), P(i16(dIdx)), P(i16(dIdy)), P(f32(last_depth)), P(f32(next_depth)), P(u8(last_image)),
                           P(u8(next_image)), P(corres), F(max_dd), P(f32(kt)), P(f32(krkinv).reshape(9)), w, h, launch[0],
                           launch[1], C.byref(sig), C.byref(cnt), P(err))
    return corres, sig.value, cnt.value, err


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobel_scale, launch=RGB_LAUNCH):
    h, w = dIdx.shape
    A = np.zeros(36, np.float32); b = np.zeros(6, np.float32)
    lib().ref_rgb_step(P(np.ascontiguousarray(corres)), F(sigma), P(f32(cloud)), F(fx), F(fy), P(i16(dIdx)), P(i16(dIdy)),
                       F(sobel_scale), w, h, launch[0], launch[1], P(A), P(b))
    return A.reshape(6, 6), b


def so3_step(last_image, next_image, basis, kinv, krlr, launch=SO3_LAUNCH):
    h, w = next_image.shape
    A = np.zeros(9, np.float32); b = np.zeros(3, np.float32); r = np.zeros(2, np.float32)
    lib().ref_so3_step(P(u8(last_image)), P(u8(next_image)), P(f32(basis).reshape(9)), P(f32(kinv).reshape(9)),
                       P(f32(krlr).reshape(9)), w, h, launch[0], launch[1], P(A), P(b), P(r))
    return A.reshape(3, 3), b, r


# ---- surfel passes: the reference's own GLSL shaders (oracle/ref_shim/ref_gl.cpp harness) -------------------------------------
class _SurfelAlias:
    """looks like tests/orc.py's `lib` to tests/orc_pipeline.py, but resolves orc_X to ref_X in the reference library
    (orc_inverse_pose / orc_fusion_weight / orc_requires_fill_in are host code in the reference, not shaders: they stay oracle)."""
    HOST = ("orc_inverse_pose", "orc_fusion_weight", "orc_requires_fill_in")

    def __getattr__(self, name):
        import orc
        if name in self.HOST:
            return getattr(orc.lib, name)
        return getattr(lib(), name.replace("orc_", "ref_", 1))


class surfel_passes:
    """context manager: inside it the functions of tests/orc_pipeline.py run the reference's shaders instead of the oracle."""

    def __enter__(self):
        import orc_pipeline
        self._mod = orc_pipeline
        self._saved = orc_pipeline.lib
        orc_pipeline.lib = _SurfelAlias()
        return orc_pipeline

    def __exit__(self, *exc):
        self._mod.lib = self._saved
        return False
