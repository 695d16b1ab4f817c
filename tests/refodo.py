"""ctypes binding of the reference's OWN RGBDOdometry class (Core/Utils/RGBDOdometry.{h,cpp} + OdometryProvider.h, compiled as they lie
into oracle/_ref/libcofusion_ref.so: CUDA kernels under the CPU SIMT emulator, Eigen / GPUTexture / Stopwatch stand-ins of
oracle/ref_shim).  Test infrastructure only.  Same method names as orc.Odometry, so that a frame's tracking inputs can be handed
to both.  Also: the recorder that collects those inputs from a run of the oracle's -static frame loop."""
from __future__ import annotations

import ctypes as C

import numpy as np

import orc
import ref
from orc import P, f32, u8

# option sets of RGBDOdometry::getIncrementalTransformation exercised by the pin (name, rgb_only, icp_weight, pyramid, fast_odom, so3)
OPTION_SETS = [("default", False, 10.0, True, False, True), ("icp_only", False, 100.0, True, False, True),
               ("rgb_only", True, 10.0, True, False, True), ("no_so3", False, 10.0, True, False, False),
               ("fast_odom", False, 10.0, True, True, True), ("no_pyramid", False, 10.0, False, False, True)]


class RefOdometry:
    def __init__(self, w, h, cx, cy, fx, fy):
        self.lib = ref.lib()
        self.lib.ref_odo_create.restype = C.c_void_p
        self.w, self.h = w, h
        self.h_ = self.lib.ref_odo_create(w, h, C.c_float(cx), C.c_float(cy), C.c_float(fx), C.c_float(fy))

    def __del__(self):
        if getattr(self, "h_", None):
            self.lib.ref_odo_destroy(C.c_void_p(self.h_))
            self.h_ = None

    def init_first_rgb(self, rgba):
        self.lib.ref_odo_init_first_rgb(C.c_void_p(self.h_), P(u8(rgba)))

    def init_icp_model(self, v4, n4, pose):
        self.lib.ref_odo_init_icp_model(C.c_void_p(self.h_), P(f32(v4)), P(f32(n4)), C.c_float(20.0), P(f32(pose).reshape(16)))

    def init_rgb_model(self, rgba):
        self.lib.ref_odo_init_rgb_model(C.c_void_p(self.h_), P(u8(rgba)))

    def init_icp(self, depth_pyr, cutoff):
        keep = [f32(d) for d in depth_pyr]
        arr = (C.c_void_p * 3)(*[d.ctypes.data for d in keep])
        self.lib.ref_odo_init_icp(C.c_void_p(self.h_), arr, C.c_float(cutoff))

    def init_rgb(self, rgba):
        self.lib.ref_odo_init_rgb(C.c_void_p(self.h_), P(u8(rgba)))

    def track(self, trans, rot, rgb_only=False, icp_weight=10.0, pyramid=True, fast_odom=False, so3=True, err_surface=None):
        trans = f32(trans).copy().reshape(3)
        rot = f32(rot).copy().reshape(9)
        stats = np.zeros(6, np.float32); A = np.zeros(36, np.float64); b = np.zeros(6, np.float64)
        self.lib.ref_odo_track(C.c_void_p(self.h_), P(trans), P(rot), int(rgb_only), C.c_float(icp_weight), int(pyramid), int(fast_odom),
                               int(so3), P(err_surface), P(stats), P(A), P(b))
        return trans, rot.reshape(3, 3), dict(last_icp_error=stats[0], last_icp_count=stats[1], last_rgb_error=stats[2],
                                              last_rgb_count=stats[3], last_so3_error=stats[4], last_so3_count=stats[5],
                                              lastA=A.reshape(6, 6), lastb=b)


class _Recorder:
    """stands in for the pipeline's orc.Odometry: forwards everything, keeps the arguments of the current frame"""

    def __init__(self, inner):
        self.inner = inner
        self.frames = []
        self.cur = {}
        self.prev_rgba = None

    def init_first_rgb(self, rgba):
        self.prev_rgba = u8(rgba).copy()
        self.inner.init_first_rgb(rgba)

    def init_icp_model(self, v4, n4, pose):
        self.cur = dict(v4=f32(v4).copy(), n4=f32(n4).copy(), pose=f32(pose).copy())
        self.inner.init_icp_model(v4, n4, pose)

    def init_rgb_model(self, rgba):
        self.cur["img"] = u8(rgba).copy()
        self.inner.init_rgb_model(rgba)

    def init_icp(self, depth_pyr, cutoff):
        self.cur["depth_pyr"] = [f32(d).copy() for d in depth_pyr]; self.cur["cutoff"] = float(cutoff)
        self.inner.init_icp(depth_pyr, cutoff)

    def init_rgb(self, rgba):
        self.cur["rgba"] = u8(rgba).copy(); self.cur["prev_rgba"] = self.prev_rgba
        self.inner.init_rgb(rgba)
        self.prev_rgba = u8(rgba).copy()

    def track(self, *a, **k):
        self.frames.append(self.cur)
        return self.inner.track(*a, **k)


def record_tracking_inputs(W, H, n_frames, conf_global=0.5, n_obj=0):
    """play the oracle's -static frame loop on the synthetic stream, return the tracking inputs of every tracked frame"""
    import orc_pipeline as op
    from co_fusion_amd import synth
    cam = synth.Camera.scaled(W, H)
    sc = synth.Scene(n_obj=n_obj)
    pipe = op.StaticPipeline(cam, conf_global=conf_global)
    rec = _Recorder(pipe.odom)
    pipe.odom = rec
    for t in range(n_frames):
        d, rgb, _, _ = sc.render(cam, t, noise=True)
        pipe.process_frame(d, synth.rgb_to_rgba(rgb))
    return cam, rec.frames


def track_once(cls, cam, W, H, fr, opts):
    """a fresh tracker of class `cls` (orc.Odometry or RefOdometry) on one recorded frame with one option set"""
    _, rgb_only, icp_weight, pyramid, fast_odom, so3 = opts
    od = cls(W, H, cam.cx, cam.cy, cam.fx, cam.fy)
    od.init_first_rgb(fr["prev_rgba"])   # lastNextImage = the previous frame's intensity pyramid (what the so3 swap leaves there)
    od.init_icp_model(fr["v4"], fr["n4"], fr["pose"]); od.init_rgb_model(fr["img"])
    od.init_icp(fr["depth_pyr"], fr["cutoff"]); od.init_rgb(fr["rgba"])
    err = np.zeros((H, W), np.float32)
    tr, rot, st = od.track(fr["pose"][:3, 3], fr["pose"][:3, :3], rgb_only=rgb_only, icp_weight=icp_weight, pyramid=pyramid,
                           fast_odom=fast_odom, so3=so3, err_surface=err)
    if not isinstance(st, dict):
        st = dict(last_icp_error=st.last_icp_error, last_icp_count=st.last_icp_count, last_rgb_error=st.last_rgb_error,
                  last_rgb_count=st.last_rgb_count, last_so3_error=st.last_so3_error, last_so3_count=st.last_so3_count,
                  lastA=np.array(st.lastA).reshape(6, 6), lastb=np.array(st.lastb))
    return tr, rot, st, err
