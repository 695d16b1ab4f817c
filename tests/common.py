"""Shared synthetic inputs for the parity tests (seeded, generated on the fly)."""
from __future__ import annotations

import functools
import warnings

import numpy as np

from co_fusion_amd import synth

warnings.filterwarnings("ignore", category=RuntimeWarning)


@functools.lru_cache(maxsize=8)
def frame_pair(width=640, height=480, n_obj=0, t0=0, t1=3, noise=False, seed=1234):
    """Two frames of the synthetic scene + the ideal model prediction of the first one."""
    cam = synth.Camera.scaled(width, height)
    sc = synth.Scene(n_obj=n_obj, seed=seed)
    d0, rgb0, l0, T0 = sc.render(cam, t0, noise=noise)
    d1, rgb1, l1, T1 = sc.render(cam, t1, noise=noise)
    v4, n4, img = synth.ideal_prediction(cam, d0, rgb0)
    return dict(cam=cam, d0=d0, rgb0=rgb0, rgba0=synth.rgb_to_rgba(rgb0), T0=T0, d1=d1, rgb1=rgb1,
                rgba1=synth.rgb_to_rgba(rgb1), T1=T1, v4=v4, n4=n4, img=img, l0=l0, l1=l1)


def perturbed_pose(seed=0, trans=0.005, rot_deg=0.5):
    """Small rigid perturbation (5 mm, 0.5 deg) used for the kernel micro-benchmarks."""
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    ang = np.deg2rad(rot_deg)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
    t = rng.normal(size=3); t = t / np.linalg.norm(t) * trans
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R.astype(np.float32)
    T[:3, 3] = t.astype(np.float32)
    return T
