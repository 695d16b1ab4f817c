"""Deterministic synthetic RGB-D stream.

The reference ships no data (its sequences are external downloads,
/root/reference/README.md:36-44), so every workload here is an analytic ray-cast
scene: the inside of a 3.6 x 2.2 x 3.4 m room (depths 0.8-2.8 m, the range in which Kinect-class
noise still leaves usable normals) with static clutter boxes, plus N moving
objects (spheres / oriented boxes) on smooth SE(3) paths, seen from a camera on a
slow Lissajous path.  Depth is metric f32, quantised to millimetres like a .klg
log (GUI/Tools/KlgLogReader.cpp:63-69); colour is RGB u8.

Pose convention matches the reference: the world frame is the first camera frame
and `pose` is T(world <- camera), 4x4 row-major float64.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np


@dataclasses.dataclass
class Camera:
    width: int = 640
    height: int = 480
    fx: float = 528.0
    fy: float = 528.0
    cx: float = 320.0
    cy: float = 240.0

    @staticmethod
    def scaled(width: int, height: int) -> "Camera":
        s = width / 640.0
        return Camera(width, height, 528.0 * s, 528.0 * s, width / 2.0, height / 2.0)


def _rot_axis(axis, ang):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def _texture(p, seed, base):
    """Smooth + checker albedo from local coordinates p[...,3]; returns float rgb in [0,1]."""
    rng = np.random.default_rng(seed)
    k = rng.uniform(5.0, 11.0, size=(3, 3))
    ph = rng.uniform(0, 6.28, size=(3,))
    s = np.sin(p @ k.T + ph)  # [...,3]
    chk = (np.floor(p[..., 0] * 4.0) + np.floor(p[..., 1] * 4.0) + np.floor(p[..., 2] * 4.0)) % 2.0
    tex = 0.55 + 0.22 * s + 0.18 * (chk[..., None] - 0.5)
    return np.clip(tex * base, 0.04, 1.0)


class Scene:
    """Static room + clutter + moving objects."""

    def __init__(self, n_obj: int = 0, seed: int = 1234, kinds: str | None = None):
        """kinds: None = spheres and boxes alternate (the scenes of rounds 1-4); "box" = textured boxes only (no rotationally symmetric
        shape: every object's pose is observable, tests/golden/make_ref_traj_golden.py: crf_two_boxes_640)"""
        rng = np.random.default_rng(seed)
        self.seed = seed
        # room interior, world frame = first camera frame (camera looks along +z, y down)
        self.room_min = np.array([-1.8, -1.2, -0.8])
        self.room_max = np.array([1.8, 1.0, 2.6])
        # static clutter: axis-aligned boxes standing on the floor (y = room_max[1])
        self.clutter = []
        for i in range(5):
            sx, sy, sz = rng.uniform(0.25, 0.6), rng.uniform(0.25, 0.7), rng.uniform(0.25, 0.5)
            cx = rng.uniform(-1.4, 1.4)
            cz = rng.uniform(1.5, 2.3)
            lo = np.array([cx - sx / 2, self.room_max[1] - sy, cz - sz / 2])
            hi = np.array([cx + sx / 2, self.room_max[1], cz + sz / 2])
            self.clutter.append((lo, hi, rng.uniform(0.5, 1.0, size=3), 100 + i))
        self.objects = []
        for i in range(n_obj):
            kind = kinds if kinds else ("sphere" if i % 2 == 0 else "box")
            size = rng.uniform(0.12, 0.2) if kind == "sphere" else rng.uniform(0.2, 0.35, size=3)
            c0 = np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.3, 0.4), rng.uniform(0.9, 1.5)])
            amp = rng.uniform(0.10, 0.25, size=3) * np.array([1.0, 0.4, 0.5])
            freq = rng.uniform(0.02, 0.04, size=3)  # rad/frame -> <= ~1.4 cm/frame
            phase = rng.uniform(0, 6.28, size=3)
            axis = rng.normal(size=3)
            rate = rng.uniform(0.004, 0.010)  # rad/frame (<= 0.6 deg)
            self.objects.append(dict(kind=kind, size=size, c0=c0, amp=amp, freq=freq, phase=phase, axis=axis,
                                     rate=rate, base=rng.uniform(0.45, 1.0, size=3), seed=200 + i))

    # -- trajectories -------------------------------------------------------
    def camera_pose(self, t: int) -> np.ndarray:
        """T(world <- camera) at frame t; identity at t = 0."""
        pos = np.array([0.25 * math.sin(0.020 * t), 0.08 * math.sin(0.031 * t), 0.15 * math.sin(0.013 * t)])
        yaw = 0.18 * math.sin(0.017 * t)
        pitch = 0.07 * math.sin(0.023 * t)
        R = _rot_axis([0, 1, 0], yaw) @ _rot_axis([1, 0, 0], pitch)
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = pos
        return T

    def object_pose(self, i: int, t: int) -> np.ndarray:
        """T(world <- object-local) at frame t."""
        o = self.objects[i]
        T = np.eye(4)
        T[:3, :3] = _rot_axis(o["axis"], o["rate"] * t)
        T[:3, 3] = o["c0"] + o["amp"] * np.sin(o["freq"] * t + o["phase"])
        return T

    # -- rendering ----------------------------------------------------------
    def render(self, cam: Camera, t: int, noise: bool = True):
        """Returns depth f32 [H,W] (m), rgb u8 [H,W,3], label u8 [H,W] (0 = static), pose 4x4."""
        H, W = cam.height, cam.width
        T = self.camera_pose(t)
        R, c = T[:3, :3], T[:3, 3]
        u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        d_cam = np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u)], axis=-1)
        d = d_cam @ R.T  # world directions, param s.t. camera z == t
        best_t = np.full((H, W), np.inf)
        color = np.zeros((H, W, 3))
        label = np.zeros((H, W), dtype=np.uint8)

        def slab(o, dd, lo, hi):
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / dd
                t0 = (lo - o) * inv
                t1 = (hi - o) * inv
            tmin = np.minimum(t0, t1).max(axis=-1)
            tmax = np.maximum(t0, t1).min(axis=-1)
            return tmin, tmax

        # room: we are inside, hit = exit point
        _, tmax = slab(c, d, self.room_min, self.room_max)
        p = c + d * tmax[..., None]
        best_t = tmax
        color = _texture(p, self.seed + 1, np.array([0.9, 0.85, 0.8]))
        # clutter
        for lo, hi, base, sd in self.clutter:
            tmin, tmax_b = slab(c, d, lo, hi)
            hit = (tmin < tmax_b) & (tmin > 0.05) & (tmin < best_t)
            if hit.any():
                p = c + d * tmin[..., None]
                col = _texture(p - lo, sd, base)
                best_t = np.where(hit, tmin, best_t)
                color = np.where(hit[..., None], col, color)
        # moving objects
        for i, o in enumerate(self.objects):
            To = self.object_pose(i, t)
            Ro, co = To[:3, :3], To[:3, 3]
            ol = (c - co) @ Ro  # camera centre in object frame
            dl = d @ Ro
            if o["kind"] == "sphere":
                r = float(o["size"])
                a = (dl * dl).sum(-1)
                b = 2.0 * (dl * ol).sum(-1)
                cc = float(ol @ ol) - r * r
                disc = b * b - 4 * a * cc
                with np.errstate(invalid="ignore"):
                    tt = (-b - np.sqrt(disc)) / (2 * a)
                hit = (disc > 0) & (tt > 0.05) & (tt < best_t)
            else:
                half = np.asarray(o["size"]) / 2.0
                tmin, tmax_b = slab(ol, dl, -half, half)
                tt = tmin
                hit = (tmin < tmax_b) & (tmin > 0.05) & (tmin < best_t)
            if hit.any():
                pl = ol + dl * np.where(hit, tt, 0.0)[..., None]
                col = _texture(pl, o["seed"], o["base"])
                best_t = np.where(hit, tt, best_t)
                color = np.where(hit[..., None], col, color)
                label = np.where(hit, np.uint8(i + 1), label)

        # simple headlight shading so that intensity varies with geometry too
        depth = best_t.copy()
        shade = np.clip(1.15 - 0.12 * depth, 0.45, 1.0)
        rgb = color * shade[..., None] * 255.0
        if noise:
            rng = np.random.default_rng(self.seed * 7919 + 1235 + t)
            sigma = 0.0012 + 0.0019 * (depth - 0.4) ** 2
            depth = depth + rng.normal(size=depth.shape) * sigma
            drop = rng.random(size=depth.shape) < 0.005
            depth = np.where(drop, 0.0, depth)
            rgb = rgb + rng.normal(size=rgb.shape) * 2.0
        depth = np.where((depth > 0.3) & (depth < 9.0), depth, 0.0)
        depth_mm = np.round(depth * 1000.0).astype(np.uint16)
        depth_f = depth_mm.astype(np.float32) * np.float32(0.001)
        rgb_u8 = np.clip(np.round(rgb), 1, 255).astype(np.uint8)
        return depth_f, rgb_u8, label, T


def rgb_to_rgba(rgb: np.ndarray) -> np.ndarray:
    """u8 [H,W,3] -> u8 [H,W,4] with alpha 255 (the GL RGBA upload of CoFusion.cpp:179)."""
    H, W, _ = rgb.shape
    out = np.empty((H, W, 4), dtype=np.uint8)
    out[..., :3] = rgb
    out[..., 3] = 255
    return out


def ideal_prediction(cam: Camera, depth: np.ndarray, rgb: np.ndarray, conf: float = 20.0):
    """A 'perfect splat' of a frame: RGBA32F vertex(+conf) / normal(+radius) maps and RGBA8 image,
    in the layout ModelProjection::combinedPredict produces (combo_splat.frag:37-65).  Used by the
    tracking-only tests/bench before the surfel pipeline is involved."""
    H, W = depth.shape
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    z = depth.astype(np.float32)
    vert = np.zeros((H, W, 4), dtype=np.float32)
    vert[..., 0] = (u - np.float32(cam.cx)) * z / np.float32(cam.fx)  # same pixel->ray convention as createVMap (cudafuncs.cu:121)
    vert[..., 1] = (v - np.float32(cam.cy)) * z / np.float32(cam.fy)
    vert[..., 2] = z
    vert[..., 3] = np.where(z > 0, np.float32(conf), np.float32(0))
    vert[z <= 0] = 0
    nrm = np.zeros((H, W, 4), dtype=np.float32)
    p = vert[..., :3]
    dx = np.zeros_like(p)
    dy = np.zeros_like(p)
    dx[:, 1:-1] = p[:, 2:] - p[:, :-2]
    dy[1:-1, :] = p[2:, :] - p[:-2, :]
    n = np.cross(dx, dy)
    ln = np.linalg.norm(n, axis=-1)
    ok = (ln > 0) & (z > 0)
    ok[:, 1:-1] &= (z[:, 2:] > 0) & (z[:, :-2] > 0)
    ok[1:-1, :] &= (z[2:, :] > 0) & (z[:-2, :] > 0)
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    n = np.where(ok[..., None], n / np.maximum(ln, 1e-20)[..., None], 0)
    # reference convention: normals point AWAY from the camera (cudafuncs.cu:181, geometry.glsl:25-36)
    flip = (n * p).sum(-1) < 0
    n = np.where(flip[..., None], -n, n)
    nrm[..., :3] = n
    nrm[..., 3] = np.where(ok, z / np.float32(cam.fx) * np.float32(1.41421356), 0)
    vert[~ok] = 0
    img = rgb_to_rgba(rgb).copy()
    img[~ok] = 0
    return vert, nrm, img
