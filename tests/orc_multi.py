"""CoFusion::processFrame (Core/CoFusion.cpp:171-524) with multiple models, restated on the CPU oracle.
Test infrastructure only: mirrors co_fusion_amd/host/CoFusion.cpp step by step."""
from __future__ import annotations

import ctypes as C

import numpy as np

import orc
import orc_pipeline as op
from orc import P, f32, lib, u8

SURFEL = 12
TIME_DELTA = 2 ** 31 // 2 - 1


class SegParams(C.Structure):
    _fields_ = [("unaryWeightError", C.c_float), ("unaryKError", C.c_float), ("unaryThresholdNew", C.c_float),
                ("weightAppearance", C.c_float), ("weightSmoothness", C.c_float), ("scaleFeaturesRGB", C.c_float),
                ("scaleFeaturesDepth", C.c_float), ("scaleFeaturesPos", C.c_float), ("minRelSizeNew", C.c_float),
                ("maxRelSizeNew", C.c_float), ("crfIterations", C.c_int)]

    @staticmethod
    def defaults():
        f = np.float32
        return SegParams(75.0, 0.0375, 5.5, 7.0, 2.0, f(1.0) / f(10.0), f(1.0) / f(0.9), f(1.0) / f(1.8), 0.015, 0.4, 10)


class SegModel(C.Structure):
    _fields_ = [("id", C.c_uint), ("superPixelCount", C.c_uint), ("avgConfidence", C.c_float), ("depthMean", C.c_float),
                ("depthStd", C.c_float), ("top", C.c_int), ("right", C.c_int), ("bottom", C.c_int), ("left", C.c_int)]


def slic(rgba):
    h, w = rgba.shape[:2]
    out = np.zeros((h, w), np.int32)
    lib.orc_slic(P(u8(rgba)), w, h, P(out))
    return out


def segment_crf(params, rgba, depth, model_ids, icp_errs, vertconfs, next_id, allow_new):
    h, w = depth.shape
    n = len(model_ids)
    ids = (C.c_uint * n)(*model_ids)
    icp_keep = [f32(a) for a in icp_errs]; vc_keep = [f32(a) for a in vertconfs]
    icp_arr = (C.c_void_p * n)(*[a.ctypes.data for a in icp_keep])
    vc_arr = (C.c_void_p * n)(*[a.ctypes.data for a in vc_keep])
    full = np.zeros((h, w), np.uint8)
    models = (SegModel * (n + 1))()
    n_out = C.c_int(); has_new = C.c_int(); rng = C.c_float()
    labels = np.zeros((h, w), np.int32)
    low = np.zeros(((h // 16), (w // 16)), np.uint8)
    lib.orc_segment_crf(C.byref(params), w, h, P(u8(rgba)), P(f32(depth)), n, ids, icp_arr, vc_arr, C.c_uint(next_id), int(allow_new),
                        P(full), models, C.byref(n_out), C.byref(has_new), C.byref(rng), P(labels), P(low))
    md = [dict(id=m.id, superPixelCount=m.superPixelCount, avgConfidence=m.avgConfidence, depthMean=m.depthMean, depthStd=m.depthStd,
               top=m.top, right=m.right, bottom=m.bottom, left=m.left) for m in models[:n_out.value]]
    return dict(full=full, modelData=md, hasNewLabel=bool(has_new.value), depthRange=rng.value, labels=labels, low=low)


def segment_gt(gt_mask, depth, model_ids, next_id, allow_new, mapping):
    h, w = depth.shape
    n = len(model_ids)
    ids = (C.c_uint * n)(*model_ids)
    full = np.zeros((h, w), np.uint8)
    models = (SegModel * (n + 1))()
    n_out = C.c_int(); has_new = C.c_int()
    lib.orc_segment_gt(P(u8(gt_mask)), P(f32(depth)), w, h, n, ids, C.c_uint(next_id), int(allow_new), P(mapping), P(full), models,
                       C.byref(n_out), C.byref(has_new))
    md = [dict(id=m.id, superPixelCount=m.superPixelCount, avgConfidence=m.avgConfidence, depthMean=m.depthMean, depthStd=m.depthStd)
          for m in models[:n_out.value]]
    return dict(full=full, modelData=md, hasNewLabel=bool(has_new.value))


class OModel:
    """Model (Core/Model/Model.h) on the oracle."""

    def __init__(self, cam, w, h, mid, conf_threshold, fill_in):
        self.cam, self.w, self.h = cam, w, h
        self.id = mid
        self.conf_threshold = np.float32(conf_threshold)
        self.fill_in_enabled = fill_in
        self.max_depth = np.float32(3.402823466e+38)
        self.pose = np.eye(4, dtype=np.float32)
        self.last_pose = np.eye(4, dtype=np.float32)
        self.surfels = np.zeros((0, SURFEL), np.float32)
        self.odom = orc.Odometry(w, h, cam.cx, cam.cy, cam.fx, cam.fy)
        self.icp_error = np.zeros((h, w), np.float32)
        self.unseen = 0
        self.pred = None
        self.fill = None
        self.index = None
        self.stats = None

    def predict_indices(self, tick):
        self.index = op.predict_indices(self.surfels, self.pose, self.cam, self.w, self.h, 20.0, tick, TIME_DELTA)

    def combined_predict(self, tick):
        self.pred = op.combined_predict(self.surfels, self.pose, self.cam, self.w, self.h, 20.0, self.conf_threshold, tick, tick, TIME_DELTA)

    def perform_fill_in(self, rgba, depth_filt, lost=False):
        if self.fill_in_enabled:
            img, vc, nr, _ = self.pred
            self.fill = op.fill_in(vc, nr, img, depth_filt, rgba, self.cam, pass_geom=lost, pass_rgb=lost)

    def fuse(self, tick, rgba, mask, depth, depth_filt, weight_mult):
        idx, vc, ct, nr = self.index
        wgt = op.fusion_weight(self.pose, self.last_pose, weight_mult)
        md = np.float32(min(np.float32(20.0), self.max_depth))
        self.surfels, self.new = op.fuse(self.surfels, idx, vc, nr, rgba, depth, depth_filt, mask, self.pose, self.cam, tick, wgt, self.id, md)

    def clean(self, tick, depth_filt, mask, outlier):
        idx, vc, ct, nr = self.index
        self.surfels = op.clean(self.surfels, self.new, idx, vc, ct, depth_filt, mask, self.pose, self.cam, tick, self.conf_threshold, outlier,
                                TIME_DELTA, self.id)


class MultiPipeline:
    def __init__(self, cam, depth_cutoff=5.0, icp_weight=10.0, conf_global=10.0, conf_object=0.01, outlier_coeff=3.0, so3=True,
                 spawn_offset=22, seg_params=None, max_models=16, reloc=False):
        self.reloc = op.Reloc(reloc)
        self.ocam = orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy)
        self.w, self.h = cam.width, cam.height
        self.depth_cutoff, self.icp_weight, self.outlier, self.so3 = depth_cutoff, icp_weight, outlier_coeff, so3
        self.conf_object = conf_object
        self.model_spawn_offset = spawn_offset
        self.max_models = min(max_models, 255)  # host/CoFusion.cpp: a new label needs a free model slot (ids are 8 bits, 255 = rejected)
        self.spawn_offset = 0
        self.seg_params = seg_params or SegParams.defaults()
        self.tick = 1
        self.next_id = 0
        self.models = []
        self.global_model = OModel(self.ocam, self.w, self.h, self._next_model_id(True), conf_global, True)
        self.models.append(self.global_model)
        self.mask = np.zeros((self.h, self.w), np.uint8)
        self.gt_mapping = np.zeros(256, np.uint8)
        self.last_seg = None

    def _next_model_id(self, assign=False):
        nxt = self.next_id
        if assign:
            while True:
                self.next_id = (self.next_id + 1) & 255
                if all(m.id != self.next_id for m in self.models):
                    break
        return nxt

    def _predict(self, rgba, depth_filt):
        for m in self.models:
            m.combined_predict(self.tick)
            m.perform_fill_in(rgba, depth_filt, self.reloc.lost)

    def _track(self, depth_filt, rgba):
        pyr = orc.depth_pyramid(depth_filt)
        for m in self.models:
            m.last_pose = m.pose.copy()
            img, vc, nr, _ = m.pred
            if m.fill_in_enabled and op.requires_fill_in(img):
                fv, fn, fi = m.fill
                m.odom.init_icp_model(fv, fn, m.pose); m.odom.init_rgb_model(fi)
            else:
                m.odom.init_icp_model(vc, nr, m.pose); m.odom.init_rgb_model(img)
            m.odom.init_icp(pyr, 20.0)
            m.odom.init_rgb(rgba)
            m.icp_error = np.zeros((self.h, self.w), np.float32) if m.icp_error is None else m.icp_error
            tr, rot, m.stats = m.odom.track(m.pose[:3, 3], m.pose[:3, :3], icp_weight=self.icp_weight, so3=self.so3, err_surface=m.icp_error)
            m.pose = np.eye(4, dtype=np.float32)
            m.pose[:3, :3] = rot; m.pose[:3, 3] = tr

    def process_frame(self, depth, rgba, gt_mask=None):
        depth_filt = op.bilateral(depth, self.depth_cutoff)
        if self.tick == 1:
            raw, n_raw = op.vertex_feedback(rgba, depth, self.ocam, self.tick, 20.0)
            filt, _ = op.vertex_feedback(rgba, depth_filt, self.ocam, self.tick, 20.0)
            self.global_model.surfels = op.model_initialise(raw, n_raw, filt)
            self.global_model.odom.init_first_rgb(rgba)
        else:
            self._track(depth_filt, rgba)
            tracking_ok = self.reloc.after_tracking(self.global_model.stats)   # CoFusion.cpp:225, 301-338 (the order against the
            # segmentation block does not matter: nothing in between reads trackingCount / lost)
            if self.spawn_offset < self.model_spawn_offset:
                self.spawn_offset += 1
            allow_new = self.spawn_offset >= self.model_spawn_offset and len(self.models) < self.max_models
            ids = [m.id for m in self.models]
            if gt_mask is not None:
                seg = segment_gt(gt_mask, depth, ids, self._next_model_id(), allow_new, self.gt_mapping)
            else:
                seg = segment_crf(self.seg_params, rgba, depth, ids, [m.icp_error for m in self.models], [m.pred[1] for m in self.models],
                                  self._next_model_id(), allow_new)
            self.last_seg = seg
            self.mask = seg["full"]
            md = seg["modelData"]
            for i, d in enumerate(md):
                d["modelIndex"] = i if i < len(self.models) else -1
            gmd = lambda d: np.float32(np.float64(d["depthMean"]) + np.float64(d["depthStd"]) * 1.2)
            new_model = None
            if seg["hasNewLabel"]:
                new_model = OModel(self.ocam, self.w, self.h, self._next_model_id(True), self.conf_object, False)
                new_model.odom.init_first_rgb(rgba)
                self.spawn_offset = 0
                new_model.max_depth = gmd(md[-1])
            for i in range(1, len(self.models)):
                self.models[i].max_depth = gmd(md[i])
            if new_model is not None:
                new_model.predict_indices(self.tick)
                new_model.fuse(self.tick, rgba, self.mask, depth, depth_filt, 100.0)
                new_model.clean(self.tick, depth_filt, self.mask, self.outlier)
                self.models.append(new_model)
            for d in md:
                if d["superPixelCount"] <= 0:
                    m = self.models[d["modelIndex"]]
                    m.unseen += 1
                    if d["id"] != 0:
                        self.models.pop(d["modelIndex"])
                        for o in md:
                            if o["modelIndex"] > d["modelIndex"]:
                                o["modelIndex"] -= 1
            for i in range(1, len(self.models)):
                m = self.models[i]
                m.conf_threshold = np.float32(min(max(m.conf_threshold, np.float32(md[i]["avgConfidence"])), np.float32(9.0)))
            self._predict(rgba, depth_filt)
            if tracking_ok and not self.reloc.lost:   # CoFusion.cpp:463
                for m in self.models: m.predict_indices(self.tick)
                for m in self.models: m.fuse(self.tick, rgba, self.mask, depth, depth_filt, 1.0)
                for m in self.models: m.predict_indices(self.tick)
                for m in self.models: m.clean(self.tick, depth_filt, self.mask, self.outlier)
        self._predict(rgba, depth_filt)
        if not self.reloc.lost:                       # CoFusion.cpp:495
            self.tick += 1
