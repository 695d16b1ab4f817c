"""Oracle-side surfel functions + a reference-order frame pipeline (CoFusion::processFrame restated
on the CPU oracle).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

import orc
from orc import P, f32, lib, u8

lib.orc_fusion_weight.restype = C.c_float
SURFEL = 12


def inverse_pose(pose):
    out = np.zeros(16, np.float32)
    lib.orc_inverse_pose(P(f32(pose).reshape(16)), P(out))
    return out.reshape(4, 4)


def bilateral(depth, max_d):
    h, w = depth.shape
    out = np.empty((h, w), np.float32)
    lib.orc_bilateral(P(f32(depth)), w, h, C.c_float(max_d), P(out))
    return out


def vertex_feedback(rgba, depth, cam, time, max_depth):
    h, w = depth.shape
    out = np.zeros((h * w, SURFEL), np.float32)
    n = lib.orc_vertex_feedback(P(u8(rgba)), P(f32(depth)), w, h, cam, time, C.c_float(max_depth), P(out))
    return out, n


def model_initialise(raw_fb, raw_count, filt_fb):
    out = np.zeros((max(raw_count, 1), SURFEL), np.float32)
    n = lib.orc_model_initialise(P(raw_fb), raw_count, P(filt_fb), P(out))
    return out[:n].copy()


def predict_indices(surfels, pose, cam, w, h, max_depth, time, time_delta):
    idx = np.zeros((h, w), np.uint32)
    vc = np.zeros((h, w, 4), np.float32); ct = np.zeros((h, w, 4), np.float32); nr = np.zeros((h, w, 4), np.float32)
    s = f32(surfels).reshape(-1, SURFEL)
    lib.orc_predict_indices(P(s), s.shape[0], P(f32(pose).reshape(16)), cam, w, h, C.c_float(max_depth), time, time_delta,
                            P(idx), P(vc), P(ct), P(nr))
    return idx, vc, ct, nr


def combined_predict(surfels, pose, cam, w, h, max_depth, conf_threshold, time, max_time, time_delta):
    img = np.zeros((h, w, 4), np.uint8); vc = np.zeros((h, w, 4), np.float32); nr = np.zeros((h, w, 4), np.float32)
    tm = np.zeros((h, w), np.uint16)
    s = f32(surfels).reshape(-1, SURFEL)
    lib.orc_combined_predict(P(s), s.shape[0], P(f32(pose).reshape(16)), cam, w, h, C.c_float(max_depth),
                             C.c_float(conf_threshold), time, max_time, time_delta, P(img), P(vc), P(nr), P(tm))
    return img, vc, nr, tm


def fill_in(pv, pn, pimg, depth, rgba, cam, pass_geom=False, pass_rgb=False):
    h, w = depth.shape
    ov = np.empty((h, w, 4), np.float32); on = np.empty((h, w, 4), np.float32); oi = np.empty((h, w, 4), np.uint8)
    lib.orc_fill_in(P(f32(pv)), P(f32(pn)), P(u8(pimg)), P(f32(depth)), P(u8(rgba)), w, h, cam, int(pass_geom), int(pass_rgb),
                    P(ov), P(on), P(oi))
    return ov, on, oi


def covariance(lastA):
    """RGBDOdometry::getCovariance (RGBDOdometry.cpp:479) of a tracker's last normal matrix (row-major 6x6 f64)"""
    a = np.ascontiguousarray(np.asarray(lastA, np.float64).reshape(36))
    out = np.zeros(36, np.float64)
    lib.orc_covariance(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out.reshape(6, 6)


class Reloc:
    """CoFusion's `reloc` switch (constructor argument, CoFusion.h:47): trackingOk / trackingCount / lost of CoFusion::processFrame
    (CoFusion.cpp:225, 301-338).  Without the fern database (closeLoops is off in Co-Fusion) lastFrameRecovery never becomes true,
    so a lost camera stays lost."""

    def __init__(self, on):
        self.on, self.lost, self.tracking_count = bool(on), False, 0

    def after_tracking(self, stats):
        """-> trackingOk of this frame (the background's tracker statistics); updates trackingCount / lost"""
        if not self.on:
            return True
        ok = float(np.float32(stats.last_icp_error)) < 1e-04           # (a float compared with the double literal; NaN: false)
        if not self.lost:
            cov = covariance(stats.lastA)
            for i in range(6):
                if cov[i, i] > 1e-04:
                    ok = False
                    break
            if not ok:
                self.tracking_count += 1
                if self.tracking_count > 10:
                    self.lost = True
            else:
                self.tracking_count = 0
        return ok


def requires_fill_in(pimg, ratio=0.75):
    h, w = pimg.shape[:2]
    return bool(lib.orc_requires_fill_in(P(u8(pimg)), w, h, C.c_float(ratio)))


def fuse(surfels, idx, vc, nr, rgba, depth_raw, depth_filt, mask, pose, cam, time, weighting, mask_id, max_depth):
    h, w = depth_raw.shape
    s = f32(surfels).reshape(-1, SURFEL)
    out = np.zeros((max(s.shape[0], 1), SURFEL), np.float32)
    new = np.zeros((h * w // 4 + 16, SURFEL), np.float32)
    n_new = C.c_int()
    lib.orc_fuse(P(s), s.shape[0], P(idx), P(f32(vc)), P(f32(nr)), P(u8(rgba)), P(f32(depth_raw)), P(f32(depth_filt)), P(u8(mask)),
                 P(f32(pose).reshape(16)), cam, w, h, time, C.c_float(weighting), mask_id, C.c_float(max_depth), P(out), P(new),
                 C.byref(n_new))
    return out[:s.shape[0]].copy(), new[:n_new.value].copy()


def clean(surfels, new, idx, vc, ct, depth_filt, mask, pose, cam, time, conf_threshold, outlier_coeff, time_delta, mask_id):
    h, w = depth_filt.shape
    s = f32(surfels).reshape(-1, SURFEL); nw = f32(new).reshape(-1, SURFEL)
    out = np.zeros((s.shape[0] + nw.shape[0] + 1, SURFEL), np.float32)
    n = lib.orc_clean(P(s), s.shape[0], P(nw), nw.shape[0], P(idx), P(f32(vc)), P(f32(ct)), P(f32(depth_filt)), P(u8(mask)),
                      P(f32(pose).reshape(16)), cam, w, h, time, C.c_float(conf_threshold), C.c_float(outlier_coeff), time_delta,
                      mask_id, P(out))
    return out[:n].copy()


def fusion_weight(pose, last_pose, mult):
    return float(lib.orc_fusion_weight(P(f32(pose).reshape(16)), P(f32(last_pose).reshape(16)), C.c_float(mult)))


class StaticPipeline:
    """CoFusion::processFrame (CoFusion.cpp:171-524) restricted to the `-static` configuration
    (enableMultipleModels = false: one background model, all-zero mask) on the CPU oracle."""

    TIME_DELTA = 2 ** 31 // 2 - 1  # openLoop: std::numeric_limits<int>::max() / 2 (MainController.cpp:328)

    def __init__(self, cam, depth_cutoff=5.0, icp_weight=10.0, conf_global=10.0, outlier_coeff=3.0, so3=True, reloc=False):
        self.reloc = Reloc(reloc)
        self.cam = orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy)
        self.w, self.h = cam.width, cam.height
        self.depth_cutoff = depth_cutoff
        self.max_depth_processed = 20.0
        self.icp_weight = icp_weight
        self.conf_threshold = conf_global
        self.outlier_coeff = outlier_coeff
        self.so3 = so3
        self.tick = 1
        self.pose = np.eye(4, dtype=np.float32)
        self.last_pose = np.eye(4, dtype=np.float32)
        self.surfels = np.zeros((0, SURFEL), np.float32)
        self.odom = orc.Odometry(self.w, self.h, cam.cx, cam.cy, cam.fx, cam.fy)
        self.mask = np.zeros((self.h, self.w), np.uint8)
        self.pred = None
        self.fill = None
        self.stats = None

    def _predict(self, rgba, depth_filt):
        # CoFusion::predict (CoFusion.cpp:533-545)
        self.pred = combined_predict(self.surfels, self.pose, self.cam, self.w, self.h, self.max_depth_processed,
                                     self.conf_threshold, self.tick, self.tick, self.TIME_DELTA)
        img, vc, nr, _ = self.pred
        self.fill = fill_in(vc, nr, img, depth_filt, rgba, self.cam, pass_geom=self.reloc.lost, pass_rgb=self.reloc.lost)

    def process_frame(self, depth, rgba, in_pose=None):
        depth_filt = bilateral(depth, self.depth_cutoff)
        if self.tick == 1:
            raw, n_raw = vertex_feedback(rgba, depth, self.cam, self.tick, self.max_depth_processed)
            filt, _ = vertex_feedback(rgba, depth_filt, self.cam, self.tick, self.max_depth_processed)
            self.surfels = model_initialise(raw, n_raw, filt)
            self.odom.init_first_rgb(rgba)
        else:
            if in_pose is None:
                self.last_pose = self.pose.copy()
                img, vc, nr, _ = self.pred
                do_fill = requires_fill_in(img)
                if do_fill:
                    fv, fn, fi = self.fill
                    self.odom.init_icp_model(fv, fn, self.pose); self.odom.init_rgb_model(fi)
                else:
                    self.odom.init_icp_model(vc, nr, self.pose); self.odom.init_rgb_model(img)
                self.odom.init_icp(orc.depth_pyramid(depth_filt), self.max_depth_processed)
                self.odom.init_rgb(rgba)
                tr, rot, self.stats = self.odom.track(self.pose[:3, 3], self.pose[:3, :3], icp_weight=self.icp_weight, so3=self.so3)
                self.pose = np.eye(4, dtype=np.float32)
                self.pose[:3, :3] = rot; self.pose[:3, 3] = tr
                tracking_ok = self.reloc.after_tracking(self.stats)   # CoFusion.cpp:225, 301-338
            else:
                self.pose = f32(in_pose).copy(); self.last_pose = self.pose.copy()
                tracking_ok = True
            self._predict(rgba, depth_filt)
            if tracking_ok and not self.reloc.lost:                   # CoFusion.cpp:463
                idx, vc, ct, nr = predict_indices(self.surfels, self.pose, self.cam, self.w, self.h, self.max_depth_processed, self.tick,
                                                  self.TIME_DELTA)
                wgt = fusion_weight(self.pose, self.last_pose, 1.0)
                self.surfels, new = fuse(self.surfels, idx, vc, nr, rgba, depth, depth_filt, self.mask, self.pose, self.cam, self.tick,
                                         wgt, 0, self.max_depth_processed)
                idx, vc, ct, nr = predict_indices(self.surfels, self.pose, self.cam, self.w, self.h, self.max_depth_processed, self.tick,
                                                  self.TIME_DELTA)
                self.surfels = clean(self.surfels, new, idx, vc, ct, depth_filt, self.mask, self.pose, self.cam, self.tick,
                                     self.conf_threshold, self.outlier_coeff, self.TIME_DELTA, 0)
        self._predict(rgba, depth_filt)
        if not self.reloc.lost:                                       # CoFusion.cpp:495
            self.tick += 1
        return self.pose.copy(), self.surfels.shape[0]
