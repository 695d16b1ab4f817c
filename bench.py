#!/usr/bin/env python
"""Headline benchmark: frames/sec of the Co-Fusion hot path at 640x480 on MI355X.

Contract (task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
A "step" is one CoFusion::processFrame: the whole per-frame hot path (bilateral filter, pyramid/map
preparation, SO3 + ICP/RGB Gauss-Newton tracking of every active model, [segmentation], splat prediction +
fill-in, index map, surfel fuse, index map, clean, prediction) over one synthetic RGB-D frame that is
already resident in HBM, driven through the C++ facade (libcofusion.so) over the C-ABI (libcofusion_hip.so).

Workloads (BASELINE.json configs):
  static       configs[1]: single static background model (`-static`), 640x480            <- default / headline
  objects4     configs[2]: 4 moving objects + background, motion-CRF segmentation on
  objects4-gt  as objects4 but with ground-truth label masks (the reference's Mask####.png input mode)

N > 1: one process per GPU (torch.distributed.run); every rank runs its own independent RGB-D stream --
the path partitions over independent streams/models without a data-path collective ("scaling": "weak").
`--parallel models` instead places the object models of ONE stream on the ranks (model-parallel frame loop with
an exact int64 all-reduce; "scaling": "strong").
Timing: barrier + synchronize on both sides of exactly K steps, MAX over ranks.

The JSON line also carries
  roofline      achieved algorithmic bytes/s of the dominant kernel -- the level-0 launch that carries the ICP
                reduction ((24 + 24*M) B/pixel for M lock-step models) together with the
                RGB residual pass (27*M B/pixel) -- from the dispatches' own begin/end timestamps (hipEvents
                attached to the launch on the launch stream); `traffic` = HBM-side bytes per launch from the
                committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes (profiles/r01_icp_traffic.json);
  cpu_baseline  the CPU oracle's restatement of the same frame loop ("port": the reference cannot be built
                here), timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="static", choices=["static", "objects4", "objects4-gt"])
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--frames", type=int, default=16, help="distinct synthetic frames (played forwards then backwards)")
    ap.add_argument("--cpu-frames", type=int, default=13, help="frames of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--icp-threads", type=int, default=256)
    ap.add_argument("--icp-ppt", type=int, default=1)
    ap.add_argument("--max-surfels", type=int, default=1 << 21)
    ap.add_argument("--parallel", default="streams", choices=["streams", "models"],
                    help="N > 1: 'streams' = one independent sequence per GPU (weak scaling, the default); 'models' = ONE sequence, "
                         "its object models placed on the GPUs (model-parallel frame loop, strong scaling; use an objects workload)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra configs[2] (4 objects + CRF) measurement of the default run")
    ap.add_argument("--streams", type=int, default=1, help="independent RGB-D streams per GPU (own context + HIP stream + host thread each); "
                    "1 = the headline single-sequence figure, >1 = throughput mode")
    return ap.parse_args(argv)


def make_stream(width, height, n_frames, n_obj=0, seed=1234):
    """Seeded synthetic RGB-D stream (co_fusion_amd/synth.py): noisy depth (mm-quantised), RGB, label masks."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    from co_fusion_amd import synth
    cam = synth.Camera.scaled(width, height)
    sc = synth.Scene(n_obj=n_obj, seed=seed)
    frames = []
    for t in range(n_frames):
        d, rgb, lab, T = sc.render(cam, t, noise=True)
        frames.append(dict(depth=d, rgb=rgb, rgba=synth.rgb_to_rgba(rgb), label=lab, T=T))
    return cam, frames


def frame_index(i, n):
    """0,1,..,n-1,n-2,..,1,0,1,.. : keeps the inter-frame motion small for any number of steps."""
    period = 2 * (n - 1)
    k = i % period
    return k if k < n else period - k


def timed_region(step_fn, steps, warmup, barrier, all_reduce_max, run_range=None):
    """The driver's timing contract: W untimed steps, barrier+sync, EXACTLY K steps, barrier+sync, MAX over ranks.
    run_range(lo, hi) (optional) executes steps lo..hi-1 itself (used for several streams per GPU)."""
    run = run_range or (lambda lo, hi: [step_fn(i) for i in range(lo, hi)])
    run(0, warmup)
    barrier()
    t0 = time.perf_counter()
    run(warmup, warmup + steps)
    barrier()
    return all_reduce_max(time.perf_counter() - t0)


def main(argv=None):
    args = parse(argv)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CF_BENCH_BACKEND", "nccl")  # "gloo": dry run of the multi-rank path on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if os.environ.get("CF_BENCH_SHARE_GPU"):  # dry run only: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    from co_fusion_amd import facade

    W, H = args.width, args.height
    n_obj = 0 if args.workload == "static" else 4
    import threading
    S = max(1, args.streams)
    use_gt = args.workload == "objects4-gt"
    dev = torch.device("cuda", local_rank)
    streams = []
    for si in range(S):
        model_parallel = args.parallel == "models" and world > 1
        cam, frames = make_stream(W, H, args.frames, n_obj=n_obj, seed=1234 + (0 if model_parallel else rank * 64) + si)
        cfi = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels,
                              enable_multiple_models=int(n_obj > 0), device_frames_complete=1,  # the ring of frames is resident before timing starts
                              **(dict(rank=rank, world=world) if model_parallel else {}))
        if model_parallel:
            cfi.set_allreduce()
        cfi.set_icp_launch(args.icp_threads, args.icp_ppt)
        hip_stream = None
        if S > 1:  # every stream of work on its own HIP stream (the default is torch's current stream)
            hip_stream = torch.cuda.Stream(device=dev)
            cfi.set_stream(hip_stream)
        resident = [dict(depth=torch.from_numpy(f["depth"]).to(dev), rgba=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
        streams.append(dict(cf=cfi, frames=frames, resident=resident, hip_stream=hip_stream))
    cf, frames = streams[0]["cf"], streams[0]["frames"]
    torch.cuda.synchronize()

    def step_stream(st, i):
        k = frame_index(i, args.frames)
        if use_gt:  # GT masks are a host-side input of the reference (FrameData.mask); depth/rgb stay host too in this mode
            f = st["frames"][k]
            st["cf"].process_frame(f["depth"], f["rgb"], mask=(f["label"] * 40).astype(np.uint8), timestamp=i)
        else:
            st["cf"].process_frame_device(st["resident"][k]["depth"], st["resident"][k]["rgba"], timestamp=i)

    def run_range(lo, hi):
        """steps lo..hi-1 of every stream; streams beyond the first run on their own host threads (ctypes drops the GIL)"""
        if S == 1:
            for i in range(lo, hi):
                step_stream(streams[0], i)
            return
        def work(st):
            torch.cuda.set_device(local_rank)  # HIP's current device is per thread
            for i in range(lo, hi):
                step_stream(st, i)
        ths = [threading.Thread(target=work, args=(st,)) for st in streams]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def all_reduce_max(dt):
        if dist is None:
            return dt
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # warm-up is untimed; profiling counters only cover the timed steps
    run_range(0, args.warmup)
    cf.profile_enable(True)
    cf.profile_read(reset=True)
    dt = timed_region(None, args.steps, 0, barrier, all_reduce_max, run_range=lambda lo, hi: run_range(lo + args.warmup, hi + args.warmup))
    prof = cf.profile_read(reset=True)
    cf.profile_enable(False)
    model_parallel = args.parallel == "models" and world > 1
    fps = args.steps * (1 if model_parallel else world) * S / dt

    out = None
    if rank == 0:
        n_models = cf.num_models
        counts = [cf.model_info(i)["count"] for i in range(n_models)]
        achieved = (prof.icp_bytes / 1e9) / (prof.icp_ms_total / 1e3) if prof.icp_ms_total > 0 else 0.0
        roofline = dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                        traffic=pmc_traffic(args.workload, W * H), kernel="cf::icp_reduce_kernel<PPT,0>: ICP reduction || RGB residual, pyramid level 0", launches=int(prof.icp_launches),
                        avg_us=round(1e3 * prof.icp_ms_total / max(1, prof.icp_launches), 3),
                        bytes_per_launch=int(prof.icp_bytes / max(1, prof.icp_launches)))
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(cam, frames, min(args.cpu_frames, args.frames), args.workload)
        desc = {"static": "single static background model (-static)", "objects4": "4 moving objects + background, motion-CRF segmentation",
                "objects4-gt": "4 moving objects + background, ground-truth label masks"}[args.workload]
        out = dict(metric="frames/sec at 640x480 (N active models) + ICP-reduce achieved HBM GB/s vs peak", value=round(fps, 2),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 4),
                   higher_is_better=True, scaling="strong" if model_parallel else "weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=f"{desc}, {W}x{H} synthetic noisy RGB-D, whole CoFusion::processFrame hot path "
                                        "(bilateral, tracking SO3+4/5/10 ICP+RGB GN, predict, fuse, clean)",
                               active_models=n_models, surfels=counts, icp_launch=[args.icp_threads, args.icp_ppt],
                               streams_per_gpu=S, parallel=args.parallel if world > 1 else "single",
                               frames="ring of device-resident frames, complete before each call (device_frames_complete=1)"),
                   roofline=roofline, cpu_baseline=cpu)
        if world == 1 and args.workload == "static" and S == 1 and not args.no_secondary:
            # BASELINE.json's target sentence is phrased on configs[2] (4 moving objects + background, CRF on): measured too
            out["secondary"] = secondary_objects4(args, torch, facade, local_rank)
        print(json.dumps(out))
    for st in streams:
        st["cf"].close()
    if dist is not None:
        dist.destroy_process_group()
    return out


def secondary_objects4(args, torch, facade, local_rank, warmup=150, steps=60):
    """configs[2]: 4 moving objects + background with the motion CRF, same timing rules (frames resident, sync on both sides)."""
    try:
        W, H = args.width, args.height
        cam, frames = make_stream(W, H, args.frames, n_obj=4, seed=1234)
        cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels, enable_multiple_models=1,
                             device_frames_complete=1)
        dev = torch.device("cuda", local_rank)
        res = [dict(depth=torch.from_numpy(f["depth"]).to(dev), rgba=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
        for i in range(warmup):
            k = frame_index(i, args.frames)
            cf.process_frame_device(res[k]["depth"], res[k]["rgba"], timestamp=i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            k = frame_index(i, args.frames)
            cf.process_frame_device(res[k]["depth"], res[k]["rgba"], timestamp=i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = cf.num_models
        counts = [cf.model_info(i)["count"] for i in range(n)]
        cf.close()
        return dict(workload="configs[2]: 4 moving objects + background, motion-CRF segmentation, synthetic", value=round(steps / dt, 2),
                    unit="frames/s", ms_per_step=round(1e3 * dt / steps, 4), warmup=warmup, steps=steps, active_models=n, surfels=counts)
    except Exception as e:  # the headline line must not depend on this extra
        return dict(error=str(e))


def pmc_traffic(workload, pixels):
    """HBM-side bytes per launch of the level-0 ICP kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE cannot be collected from inside this process); null when no pass matches this workload/shape."""
    path = os.path.join(ROOT, "profiles", "r01_icp_traffic.json")
    try:
        t = json.load(open(path))
    except OSError:
        return None
    if t.get("workload") != workload or t.get("pixels") != pixels:
        return None
    return int(t["traffic_bytes_per_launch"])


def cpu_baseline(cam, frames, n, workload):
    """CPU oracle frame loop (port of the reference path) on a bounded sample of the same stream, one host thread."""
    import ctypes
    import orc_multi as om
    import orc_pipeline as op
    threads = 1
    try:  # the oracle's only parallel loop is the bilateral filter (OpenMP): pin it to one thread so that `cores` is exact
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(1)
    except OSError:
        threads = os.cpu_count()
    if workload == "static":
        pipe = op.StaticPipeline(cam)
        run = lambda f: pipe.process_frame(f["depth"], f["rgba"])
    else:
        pipe = om.MultiPipeline(cam)
        gt = workload == "objects4-gt"
        run = lambda f: pipe.process_frame(f["depth"], f["rgba"], gt_mask=(f["label"] * 40).astype(np.uint8) if gt else None)
    run(frames[0])  # bootstrap frame (no tracking), untimed
    t0 = time.perf_counter()
    for k in range(1, n):
        run(frames[k])
    dt = time.perf_counter() - t0
    return dict(value=round((n - 1) / dt, 3), unit="frames/s", cores=threads, kind="port",
                sample=f"{n - 1} frames of the same workload ({dt:.1f} s); C oracle (gcc -O2) restating the reference frame loop, "
                       f"{threads} host thread(s) of {os.cpu_count()}")


if __name__ == "__main__":
    main()
