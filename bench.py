#!/usr/bin/env python
"""Headline benchmark: frames/sec of the Co-Fusion hot path at 640x480 on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
from rank 0.  A "step" is one pass of the hot path over one synthetic RGB-D frame that is already
resident in HBM.  At N>1 the driver launches one process per GPU through torch.distributed.run.

The JSON line also carries
  roofline      achieved algorithmic bytes/s of the dominant kernel (ICP reduction, 48 B/pixel/launch,
                BASELINE.md section 3) measured live with hipEvents on the launch stream, vs HBM peak;
  cpu_baseline  the CPU oracle's odometry path (a port of the reference's algorithm; the reference itself
                cannot be built here) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="static", choices=["static"])
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames cycled through")
    ap.add_argument("--cpu-frames", type=int, default=12, help="frames of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--icp-threads", type=int, default=256)
    ap.add_argument("--icp-ppt", type=int, default=1)
    return ap.parse_args()


def make_stream(width, height, n_frames, n_obj=0):
    """Synthetic 640x480 RGB-D stream + the model prediction each frame is tracked against."""
    import warnings
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    from co_fusion_amd import synth
    cam = synth.Camera.scaled(width, height)
    sc = synth.Scene(n_obj=n_obj, seed=1234)
    frames = []
    for t in range(n_frames + 1):
        d, rgb, _, T = sc.render(cam, t, noise=False)
        v4, n4, img = synth.ideal_prediction(cam, d, rgb)
        frames.append(dict(depth=d, rgba=synth.rgb_to_rgba(rgb), T=T, v4=v4, n4=n4, img=img))
    return cam, frames


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    from co_fusion_amd import api

    W, H = args.width, args.height
    cam, frames = make_stream(W, H, args.frames)
    ctx = api.Context(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank)
    ctx.set_icp_launch(args.icp_threads, args.icp_ppt)
    od = api.Odometry(ctx)
    dev = [dict(depth=ctx.to_device(f["depth"]), rgba=ctx.to_device(f["rgba"]), v4=ctx.to_device(f["v4"]),
                n4=ctx.to_device(f["n4"]), img=ctx.to_device(f["img"])) for f in frames]
    poses = [f["T"].astype(np.float32) for f in frames]
    od.init_first_rgb(dev[0]["rgba"])
    torch.cuda.synchronize()

    def step(i):
        """Track frame k+1 against the model prediction of frame k (frame-to-model odometry)."""
        k = i % args.frames
        prev, cur = dev[k], dev[k + 1]
        od.init_icp_model(prev["v4"], prev["n4"], poses[k])
        od.init_rgb_model(prev["img"])
        od.init_icp(ctx.depth_pyramid(cur["depth"]), 20.0)
        od.init_rgb(cur["rgba"])
        return od.track(poses[k][:3, 3], poses[k][:3, :3])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)

    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    fps = args.steps * world / dt

    out = None
    if rank == 0:
        achieved = (prof.icp_bytes / 1e9) / (prof.icp_ms_total / 1e3) if prof.icp_ms_total > 0 else 0.0
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None, kernel="icp_reduce_kernel",
                        launches=int(prof.icp_launches), avg_us=round(1e3 * prof.icp_ms_total / max(1, prof.icp_launches), 3),
                        bytes_per_launch_avg=int(prof.icp_bytes / max(1, prof.icp_launches)))
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(cam, frames, min(args.cpu_frames, args.frames))
        out = dict(metric="frames/sec at 640x480 (N active models) + ICP-reduce achieved HBM GB/s vs peak",
                   value=round(fps, 2), unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(1e3 * dt / args.steps, 4), higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=f"single static background model, {W}x{H} synthetic RGB-D, "
                                        "frame-to-model tracking (SO3 + 4/5/10 ICP+RGB Gauss-Newton)",
                               active_models=1, icp_launch=[args.icp_threads, args.icp_ppt]),
                   roofline=roofline, cpu_baseline=cpu)
        print(json.dumps(out))
    od.close()
    ctx.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return out


def cpu_baseline(cam, frames, n):
    """CPU oracle odometry (port of the reference path; 1 thread) on a bounded sample of the stream."""
    import orc
    od = orc.Odometry(cam.width, cam.height, cam.cx, cam.cy, cam.fx, cam.fy)
    od.init_first_rgb(frames[0]["rgba"])
    t0 = time.perf_counter()
    for k in range(n):
        prev, cur = frames[k], frames[k + 1]
        pose = prev["T"].astype(np.float32)
        od.init_icp_model(prev["v4"], prev["n4"], pose)
        od.init_rgb_model(prev["img"])
        od.init_icp(orc.depth_pyramid(cur["depth"]), 20.0)
        od.init_rgb(cur["rgba"])
        od.track(pose[:3, 3], pose[:3, :3])
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 3), unit="frames/s", cores=1, kind="port",
                sample=f"{n} frames of the same workload, single-thread C oracle (gcc -O2)")


if __name__ == "__main__":
    main()
