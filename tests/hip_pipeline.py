"""Test scaffolding: the reference's `-static` frame loop driven call by call over the Python mirrors of the C-ABI objects
(co_fusion_amd.api.Odometry, co_fusion_amd.model.Model) -- lock-step parity of every intermediate buffer against the oracle pipeline.
The product's frame loop is the C++ facade (co_fusion_amd/host/CoFusion.cpp); this class is not part of the package."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from co_fusion_amd.api import Context, Odometry, _f, _p
from co_fusion_amd.model import SURFEL, TIME_DELTA, Model, _DevView, bilateral, fusion_weight


class StaticPipeline:
    """CoFusion::processFrame (Core/CoFusion.cpp:171-524) for the `-static` configuration (one background
    model, all-zero mask), driven from Python over the C-ABI.  Mirrors tests/orc_pipeline.StaticPipeline."""

    def __init__(self, ctx: Context, max_surfels=1 << 20, depth_cutoff=5.0, icp_weight=10.0, conf_global=10.0, outlier_coeff=3.0,
                 so3=True):
        self.ctx = ctx
        self.model = Model(ctx, max_surfels)
        self.odom = Odometry(ctx)
        self.depth_cutoff = depth_cutoff
        self.max_depth_processed = 20.0
        self.icp_weight = icp_weight
        self.conf_threshold = conf_global
        self.outlier_coeff = outlier_coeff
        self.so3 = so3
        self.tick = 1
        self.pose = np.eye(4, dtype=np.float32)
        self.last_pose = np.eye(4, dtype=np.float32)
        self.mask = torch.zeros((ctx.height, ctx.width), dtype=torch.uint8, device=ctx.device)
        self.stats = None
        self.sync_pose = None  # test hook: callable(frame_index, pose) -> pose to continue with

    def view(self, which):
        return _DevView(self.model.tensor(which)[0])

    def _predict(self, rgba, depth_filt):
        m = self.model
        m.combined_predict(self.pose, self.max_depth_processed, self.conf_threshold, self.tick, self.tick)
        m.perform_fill_in(rgba, depth_filt)

    def process_frame(self, depth, rgba, in_pose=None):
        ctx, m = self.ctx, self.model
        depth_filt = bilateral(ctx, depth, self.depth_cutoff)
        if self.tick == 1:
            m.initialise(rgba, depth, depth_filt, self.tick, self.max_depth_processed)
            self.odom.init_first_rgb(rgba)
        else:
            if in_pose is None:
                self.last_pose = self.pose.copy()
                if m.requires_fill_in():
                    self.odom.init_icp_model(self.view(8), self.view(9), self.pose); self.odom.init_rgb_model(self.view(10))
                else:
                    self.odom.init_icp_model(self.view(5), self.view(6), self.pose); self.odom.init_rgb_model(self.view(4))
                self.odom.init_icp(ctx.depth_pyramid(depth_filt), self.max_depth_processed)
                self.odom.init_rgb(rgba)
                tr, rot, self.stats = self.odom.track(self.pose[:3, 3], self.pose[:3, :3], icp_weight=self.icp_weight, so3=self.so3)
                self.pose = np.eye(4, dtype=np.float32)
                self.pose[:3, :3] = rot; self.pose[:3, 3] = tr
                if self.sync_pose is not None:
                    self.pose = np.asarray(self.sync_pose(self.tick, self.pose), np.float32)
            else:
                self.pose = np.asarray(in_pose, np.float32).copy(); self.last_pose = self.pose.copy()
            self._predict(rgba, depth_filt)
            m.predict_indices(self.pose, self.tick, self.max_depth_processed)
            wgt = fusion_weight(ctx, self.pose, self.last_pose, 1.0)
            m.fuse(self.pose, self.tick, rgba, self.mask, depth, depth_filt, self.max_depth_processed, wgt, 0)
            m.predict_indices(self.pose, self.tick, self.max_depth_processed)
            m.clean(self.pose, self.tick, self.conf_threshold, self.outlier_coeff, depth_filt, self.mask, 0)
        self._predict(rgba, depth_filt)
        self.tick += 1
        return self.pose.copy(), m.count()

    def close(self):
        self.odom.close()
        self.model.close()
