"""ctypes binding of the reference's OWN frame loop -- CoFusion::processFrame and its helpers, cut out of Core/CoFusion.cpp at build
time and compiled into oracle/_ref/libcofusion_ref.so behind oracle/ref_shim/stub/CoFusionPin.h (every pass of a frame runs on the
CPU oracle; Core/Segmentation is the reference's own as well).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

import ref
from orc import P, f32, u8


class RefCoFusion:
    def __init__(self, cam, conf_global=10.0, conf_object=0.01, depth_cutoff=5.0, icp_weight=10.0, so3=True, spawn_offset=20, multi=True,
                 rgb_only=False, pyramid=True, fast_odom=False, frame_to_frame_rgb=False, reference_tracker=False, reloc=False):
        """reference_tracker: every model tracks with the reference's own RGBDOdometry class (its CUDA kernels under the emulator, f32 tree
        reductions, Core/Utils/RGBDOdometry.cpp:217-477) instead of the oracle's restatement with exact integer sums"""
        self.lib = ref.lib()
        self.lib.ref_cf_use_reference_tracker(int(reference_tracker))
        self.lib.ref_cf_set_reloc(int(reloc))   # the constructor's `reloc` argument (CoFusion.h:47; -rl)
        self.lib.ref_cf_create.restype = C.c_void_p
        self.w, self.h = cam.width, cam.height
        self.h_ = self.lib.ref_cf_create(self.w, self.h, C.c_float(cam.fx), C.c_float(cam.fy), C.c_float(cam.cx), C.c_float(cam.cy),
                                         C.c_float(conf_global), C.c_float(conf_object), C.c_float(depth_cutoff), C.c_float(icp_weight),
                                         int(so3), C.c_uint(spawn_offset), int(multi))
        self.lib.ref_cf_set_tracking_options(C.c_void_p(self.h_), int(rgb_only), int(pyramid), int(fast_odom), int(frame_to_frame_rgb))

    def __del__(self):
        if getattr(self, "h_", None):
            self.lib.ref_cf_destroy(C.c_void_p(self.h_))
            self.h_ = None

    def process_frame(self, depth, rgb, gt_mask=None, timestamp=0, in_pose=None):
        self.lib.ref_cf_process_frame_pose(C.c_void_p(self.h_), P(f32(depth)), P(u8(rgb)), P(u8(gt_mask)) if gt_mask is not None else None,
                                           C.c_longlong(timestamp), P(f32(in_pose).reshape(16)) if in_pose is not None else None)

    @property
    def num_models(self):
        return self.lib.ref_cf_num_models(C.c_void_p(self.h_))

    @property
    def tick(self):
        return self.lib.ref_cf_tick(C.c_void_p(self.h_))

    @property
    def lost(self):
        return bool(self.lib.ref_cf_lost(C.c_void_p(self.h_)))

    def model(self, i):
        mid = C.c_uint(); conf = C.c_float(); unseen = C.c_uint(); nlog = C.c_int()
        pose = np.zeros(16, np.float32)
        n = self.lib.ref_cf_model_info(C.c_void_p(self.h_), i, C.byref(mid), P(pose), C.byref(conf), C.byref(unseen), C.byref(nlog))
        s = np.zeros((n, 12), np.float32)
        if n:
            self.lib.ref_cf_model_surfels(C.c_void_p(self.h_), i, P(s))
        log = np.zeros(7, np.float32)
        self.lib.ref_cf_model_last_pose_log(C.c_void_p(self.h_), i, P(log))
        return dict(id=mid.value, pose=pose.reshape(4, 4), conf_threshold=np.float32(conf.value), unseen=unseen.value, count=n, surfels=s,
                    pose_log_items=nlog.value, last_pose_log=log)

    def mask(self):
        m = np.zeros((self.h, self.w), np.uint8)
        self.lib.ref_cf_mask(C.c_void_p(self.h_), P(m))
        return m

    def gl_draws(self):
        """draw calls of the pasted Model::fuse / Model::clean text served so far in this process: (data, update, unstable)"""
        d = (C.c_uint * 3)()
        self.lib.ref_cf_glpin_draws(d)
        return tuple(int(x) for x in d)
