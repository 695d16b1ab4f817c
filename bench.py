#!/usr/bin/env python
"""Headline benchmark: frames/sec of the Co-Fusion hot path at 640x480 on MI355X.

Contract (task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line from rank 0.
A "step" is one CoFusion::processFrame: the whole per-frame hot path (bilateral filter, pyramid/map
preparation, SO3 + ICP/RGB Gauss-Newton tracking of every active model, motion segmentation, splat prediction +
fill-in, index map, surfel fuse, index map, clean, prediction) over one synthetic RGB-D frame, driven through the
C++ facade (libcofusion.so) over the C-ABI (libcofusion_hip.so).

Workloads (BASELINE.json configs):
  objects4     configs[2]: 4 moving objects + background, motion-CRF segmentation on, 640x480     <- default / headline
  static       configs[1]: single static background model (`-static`), 640x480                      (reported as "secondary")
  objects4-gt  as objects4 but with ground-truth label masks (the reference's Mask####.png input mode)
  objects8     configs[3]: 8 moving objects + background
  big          configs[4]: 1280x960, 4 objects, 32 M surfels per model

Phases of one run (only the third is timed for `value`):
  pre-roll   the sequence is played until the object models exist (the motion CRF spawns at most one model every 22
             frames, CoFusion.cpp:254-262) -- building the state the metric is quoted on ("N active models");
  warm-up    W untimed steps;
  timed      barrier + synchronize, EXACTLY K steps with the frames already resident in HBM, barrier + synchronize,
             MAX over ranks;
  extras     (rank 0, N = 1) the same stream through the host-input entry point (`input: host` rate, PCIe inclusive),
             trajectory error (ATE) against the synthetic ground truth and against the CPU oracle, the CPU baseline.

N > 1: one process per GPU, the SAME workload as N = 1 (configs[2] unless --workload says otherwise; the same pre-roll with
ground-truth masks, so the same models exist at every N).  `--gpus N` without RANK in the environment re-executes itself under
torch.distributed.run.  Default partition (`--parallel models`): ONE RGB-D stream, the object models placed on the ranks
(background on rank 0), frames broadcast from the ingest GPU, poses / segmentation sums exchanged with an exact integer
all-reduce ("scaling": "strong") -- through the library's OWN RCCL communicator (cofusion_init_rccl: ncclBroadcast / ncclAllReduce
in place on the context's stream, no Python in the frame loop) when the process group is RCCL; `--collectives torch` keeps the
torch.distributed callbacks (the gloo dry runs).  `--parallel streams`: one independent sequence per GPU, no data-path
collective ("weak").

The JSON line also carries
  roofline      achieved algorithmic bytes/s of the dominant kernel -- the level-0 launch that carries the ICP reduction of
                all M lock-step models ((24 + 24*M) B/pixel) together with their RGB residual passes (11*M
                B/pixel: candidate mask, next depth + intensity, gathered last depth + intensity; the compact
                correspondence list that replaces the reference's 16 B DataTerm record is not counted) -- from the
                dispatches' own begin/end timestamps (hipEvents attached to the launch on the launch stream);
                `traffic` = HBM-side bytes per launch from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes;
  cpu_baseline  the CPU oracle's odometry path (map preparation + SO3 + 4/5/10 ICP+RGB Gauss-Newton iterations of every
                active model; "port": the reference cannot be built here) compiled -O3 -march=native -fopenmp on this box,
                timed with 1 thread (`cpu_baseline`) and with all cores (`cpu_baseline_all_cores`) on frames sampled
                from the same run (inputs = what the GPU tracker read), median per frame.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    "static": dict(n_obj=0, size=(640, 480), config="configs[1]", desc="single static background model (-static)"),
    "objects4": dict(n_obj=4, size=(640, 480), config="configs[2]", desc="4 moving objects + background, motion-CRF segmentation on"),
    "objects4-gt": dict(n_obj=4, size=(640, 480), config="configs[2] with ground-truth masks",
                        desc="4 moving objects + background, ground-truth label masks"),
    "objects8": dict(n_obj=8, size=(640, 480), config="configs[3]", desc="8 moving objects + background, motion-CRF segmentation on"),
    "big-static": dict(n_obj=0, size=(1280, 960), config="configs[4]'s frame size, one model", desc="1280x960, single static background model",
                       max_surfels=1 << 23),
    "big": dict(n_obj=4, size=(1280, 960), config="configs[4]", desc="1280x960, 4 moving objects + background, 32 M surfels per model",
                max_surfels=1 << 25),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--frames", type=int, default=16, help="distinct synthetic frames (played forwards then backwards)")
    ap.add_argument("--preroll", type=int, default=None, help="untimed frames played before the warm-up so that the object models exist "
                    "(default: 24 per object + 30 for object workloads, 10 for static)")
    ap.add_argument("--preroll-masks", default="gt", choices=["gt", "crf"],
                    help="object workloads: how the object models are spawned during the pre-roll -- 'gt': ground-truth label masks (every "
                         "object gets a model), 'crf': the motion CRF (spawns what it detects); the warm-up and the timed steps always use "
                         "the workload's own segmentation")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip host-input rate / ATE / secondary legs")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work per CPU-baseline leg")
    ap.add_argument("--icp-threads", type=int, default=256)
    ap.add_argument("--icp-ppt", type=int, default=0, help="pixels per lane of the ICP reduction of unculled trackers (0: library default = 2 at pyramid level 0, 1 below)")
    ap.add_argument("--icp-arith", default=None, choices=["product", "gram", "reference"],
                    help="rounding specification of the tracker's sums (cf_set_icp_arith; default: the library's): products rounded once / row entries "
                         "rounded once and contracted on the matrix cores / the reference's own f32 trees and host loop (a parity mode: slow by "
                         "construction); the oracle legs follow")
    ap.add_argument("--gn-mode", type=int, default=-1, help="-1: library default; 0: three launches per GN iteration; 1: two")
    ap.add_argument("--max-surfels", type=int, default=None)
    ap.add_argument("--late-index-maps", action="store_true", help="A/B: rasterise the index maps after the frame's host wait (round 4) instead of before it")
    ap.add_argument("--enqueue-threads", type=int, default=None, help="host threads enqueueing the per-model surfel passes (library default: 0)")
    ap.add_argument("--parallel", default=None, choices=["streams", "models"],
                    help="N > 1: 'models' (default for object workloads) = ONE sequence, its object models placed on the GPUs (strong "
                         "scaling); 'streams' = one independent sequence per GPU (weak scaling)")
    ap.add_argument("--shard-background", action="store_true",
                    help="N > 1, --parallel models: every rank keeps a replica of the background map and takes a share of its index-map "
                         "rasterisation (surfel range, MIN all-reduce of the z-keys) and of its ICP reduction (image rows, SUM all-reduce of the "
                         "6x6 accumulators after every launch of the Gauss-Newton loop) -- the split BASELINE.json's configs[4] names")
    ap.add_argument("--event-sampling", type=int, default=4, help="timing events on the level-0 launches of every N-th timed step (1: every step)")
    ap.add_argument("--no-kernel-events", action="store_true", help="diagnostic: do not attach timing events to the level-0 launches (no roofline figure)")
    ap.add_argument("--dry-run", action="store_true", help="plumbing test without a GPU (tests/test_cpu_distributed.py): the process-group "
                    "set-up, the timing contract and the JSON line with a stub step instead of processFrame")
    ap.add_argument("--collectives", default="auto", choices=["auto", "rccl", "torch"],
                    help="N > 1, --parallel models: 'rccl' = the library's own RCCL communicator (cofusion_init_rccl), 'torch' = "
                         "torch.distributed callbacks, 'auto' = rccl when the process group's backend is nccl")
    ap.add_argument("--lockstep", action="store_true",
                    help="--streams S > 1: the S sequences form ONE lock-step group (cofusion_group_*: one context, one set of tracking "
                         "launches for the trackers of all sequences) instead of S contexts driven by S host threads")
    ap.add_argument("--groups", type=int, default=1,
                    help="--lockstep: split the S sequences into this many lock-step groups, each with its own context, HIP stream and host "
                         "thread (the latency-bound stretches of one group run beside the throughput-bound ones of another)")
    ap.add_argument("--streams", type=int, default=1, help="independent RGB-D streams per GPU (own context + HIP stream + host thread each); "
                    "1 = the headline single-sequence figure, >1 = throughput mode")
    a = ap.parse_args(argv)
    if a.icp_arith:
        os.environ["CF_ICP_ARITH"] = a.icp_arith   # read by every context this process creates (cf_create)
    else:
        a.icp_arith = {"1": "gram", "gram": "gram", "2": "reference", "reference": "reference"}.get(os.environ.get("CF_ICP_ARITH", ""), "product")
    a.workload_defaulted = a.workload is None
    if a.workload is None:
        # the metric's configuration (configs[2]: background + 4 objects) while its five models can occupy the GPUs; beyond that BASELINE.json's
        # own multi-GPU configuration, configs[3]: "8 object models sharded one-per-GPU across 8xMI355X" (VERDICT r3 item 6).  The JSON
        # line names the workload it ran; `replicas` in the same line is the workload-independent weak-scaling figure.
        a.workload = "objects8" if a.gpus > 5 else "objects4"
    wl = WORKLOADS[a.workload]
    if a.width is None:
        a.width = wl["size"][0]
    if a.height is None:
        a.height = wl["size"][1]
    if a.max_surfels is None:
        a.max_surfels = wl.get("max_surfels", 1 << 21)
    if a.preroll is None:
        a.preroll = 10 if wl["n_obj"] == 0 else 24 * wl["n_obj"] + 30
    if a.parallel is None:
        a.parallel = "models" if wl["n_obj"] > 0 else "streams"
    return a


def make_stream(width, height, n_frames, n_obj=0, seed=1234):
    """Seeded synthetic RGB-D stream (co_fusion_amd/synth.py): noisy depth (mm-quantised), RGB, label masks, GT camera poses."""
    warnings.filterwarnings("ignore", category=RuntimeWarning)
    from co_fusion_amd import synth
    cam = synth.Camera.scaled(width, height)
    # the analytic ray caster needs ~0.7 s per 640x480 frame: rendered frames are kept in a scratch cache between runs
    cache = os.path.join(os.environ.get("CF_BENCH_CACHE", "/tmp/cf_bench_cache"), f"s{seed}_{width}x{height}_o{n_obj}_n{n_frames}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            return cam, [dict(depth=z["depth"][t], rgb=z["rgb"][t], rgba=synth.rgb_to_rgba(z["rgb"][t]), label=z["label"][t], T=z["T"][t])
                         for t in range(n_frames)]
        except Exception:  # noqa: BLE001 -- a torn cache file is simply regenerated
            pass
    sc = synth.Scene(n_obj=n_obj, seed=seed)
    frames = []
    for t in range(n_frames):
        d, rgb, lab, T = sc.render(cam, t, noise=True)
        frames.append(dict(depth=d, rgb=rgb, rgba=synth.rgb_to_rgba(rgb), label=lab, T=T))
    try:
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        tmp = cache + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, depth=np.stack([f["depth"] for f in frames]), rgb=np.stack([f["rgb"] for f in frames]),
                 label=np.stack([f["label"] for f in frames]), T=np.stack([f["T"] for f in frames]))
        os.replace(tmp, cache)
    except OSError:
        pass
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


def respawn_distributed(args, argv):
    """`--gpus N` from a plain shell: one process per GPU under torch.distributed.run (what the driver does itself)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv if argv is not None else sys.argv[1:])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    args = parse(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        rc = respawn_distributed(args, argv)
        if rc != 0:
            raise SystemExit(rc)
        return None
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    backend = None
    out_stream = sys.stdout
    if world > 1:
        # stdout carries ONE JSON line: RCCL prints a version banner when a communicator is created (torch's and the library's), so for
        # the rest of the run file descriptor 1 is routed to stderr and the line goes to a duplicate of the original descriptor
        sys.stdout.flush()
        out_stream = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("CF_BENCH_BACKEND", "nccl")  # "gloo": dry run of the multi-rank path on a 1-GPU box
        if os.environ.get("CF_BENCH_SHARE_GPU"):  # dry run only: every rank on device 0
            local_rank = 0
        if args.dry_run:
            backend = "gloo"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.dry_run:
        return dry_run(args, rank, world, dist, out_stream)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU {local_rank} (visible: {torch.cuda.device_count()})")
    torch.cuda.set_device(local_rank)
    from co_fusion_amd import facade

    W, H = args.width, args.height
    wl = WORKLOADS[args.workload]
    n_obj = wl["n_obj"]
    import threading
    S = max(1, args.streams)
    use_gt = args.workload == "objects4-gt"
    dev = torch.device("cuda", local_rank)
    model_parallel = args.parallel == "models" and world > 1 and n_obj > 0
    use_rccl = model_parallel and (args.collectives == "rccl" or (args.collectives == "auto" and backend == "nccl"))
    if args.lockstep and S > 1:
        if model_parallel or world > 1:
            raise SystemExit("bench.py: --lockstep is a single-GPU mode")
        return lockstep_run(args, torch, facade, local_rank, wl, S)
    streams = []
    for si in range(S):
        cam, frames = make_stream(W, H, args.frames, n_obj=n_obj, seed=1234 + (0 if model_parallel else rank * 64) + si)
        cfi = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels,
                              enable_multiple_models=int(n_obj > 0),
                              # single GPU / independent streams: the ring of frames is resident before timing starts; model-parallel:
                              # ranks > 0 receive every frame by broadcast just before the call, so frames are consumed in stream order
                              device_frames_complete=0 if model_parallel else 1,
                              **(dict(early_index_maps=0) if args.late_index_maps else {}),
                              **(dict(enqueue_threads=args.enqueue_threads) if args.enqueue_threads is not None else {}),
                              **(dict(rank=rank, world=world, shard_background=int(args.shard_background),
                                      colocate_background=int(n_obj >= world)) if model_parallel else {}))
        if model_parallel:
            if use_rccl:
                # the library's own ncclComm_t: every collective of the frame loop runs inside the library.  All ranks agree on whether
                # that worked; if it failed anywhere, every rank falls back to the torch.distributed callbacks
                ok = 1
                try:
                    cfi.init_rccl()
                except Exception as e:  # noqa: BLE001
                    print(f"[bench rank {rank}] cofusion_init_rccl failed: {e}", file=sys.stderr, flush=True)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    use_rccl = False
            if not use_rccl:
                cfi.set_allreduce()   # torch.distributed callbacks (process groups that are not RCCL)
        if args.icp_ppt:
            cfi.set_icp_launch(args.icp_threads, args.icp_ppt)
        if args.gn_mode >= 0:
            cfi.set_gn_mode(args.gn_mode)
        hip_stream = None
        if S > 1:  # every stream of work on its own HIP stream (the default is torch's current stream)
            hip_stream = torch.cuda.Stream(device=dev)
            cfi.set_stream(hip_stream)
        def packed(f=None):
            # depth (f32) and colour (RGBA8) of one frame in ONE allocation: the frame crosses the ranks as a single broadcast
            buf = torch.empty(H * W * 8, dtype=torch.uint8, device=dev)
            d = dict(pack=buf, depth=buf[:H * W * 4].view(torch.float32).view(H, W), rgba=buf[H * W * 4:].view(H, W, 4))
            if f is not None:
                d["depth"].copy_(torch.from_numpy(f["depth"])); d["rgba"].copy_(torch.from_numpy(f["rgba"]))
            return d
        if model_parallel and rank != 0:
            # the ingest GPU is rank 0: the other ranks own no frames, only a two-deep landing buffer for the broadcast
            resident = [packed() for _ in range(2)]
        elif model_parallel:
            resident = [packed(f) for f in frames]
        else:
            resident = [dict(depth=torch.from_numpy(f["depth"]).to(dev), rgba=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
        streams.append(dict(cf=cfi, frames=frames, resident=resident, hip_stream=hip_stream))
    cf, frames = streams[0]["cf"], streams[0]["frames"]
    torch.cuda.synchronize()

    def gt_mask(f):
        return (f["label"] * 40).astype(np.uint8)

    def step_stream(st, i, masks=None):
        k = frame_index(i, args.frames)
        if masks == "gt" or (masks is None and use_gt):
            # GT masks are a host-side input of the reference (FrameData.mask); depth/rgb stay host too in this mode
            f = st["frames"][k]
            st["cf"].process_frame(f["depth"], f["rgb"], mask=gt_mask(f), timestamp=i)
        elif model_parallel:
            # frame from the ingest GPU (rank 0) to every rank: one broadcast over xGMI (depth 1.2 MB + colour 1.2 MB at 640x480)
            buf = st["resident"][k] if rank == 0 else st["resident"][i & 1]
            if use_rccl:
                st["cf"].broadcast(buf["pack"], 0)   # ncclBroadcast on the context's stream, consumed in stream order
            else:
                dist.broadcast(buf["pack"], src=0)
            st["cf"].process_frame_device(buf["depth"], buf["rgba"], timestamp=i)
        else:
            st["cf"].process_frame_device(st["resident"][k]["depth"], st["resident"][k]["rgba"], timestamp=i)

    def run_range(lo, hi, masks=None):
        """steps lo..hi-1 of every stream; streams beyond the first run on their own host threads (ctypes drops the GIL)"""
        if S == 1:
            for i in range(lo, hi):
                step_stream(streams[0], i, masks)
            return
        def work(st):
            torch.cuda.set_device(local_rank)  # HIP's current device is per thread
            for i in range(lo, hi):
                step_stream(st, i, masks)
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

    # pre-roll (state the metric is quoted on: the object models exist), then W warm-up steps; all untimed
    P = args.preroll
    # (model-parallel too: every rank holds the host copy of the stream, ground-truth-mask frames take the host-input path on every
    # rank, so all N see the same models as N = 1)
    pre_masks = "gt" if (n_obj > 0 and args.preroll_masks == "gt") else None
    run_range(0, P, pre_masks)
    run_range(P, P + args.warmup)
    base = P + args.warmup
    cf.profile_enable(0 if args.no_kernel_events else args.event_sampling)
    cf.profile_read(reset=True)
    dt = timed_region(None, args.steps, 0, barrier, all_reduce_max, run_range=lambda lo, hi: run_range(lo + base, hi + base))
    prof = cf.profile_read(reset=True)
    cf.profile_enable(False)
    fps = args.steps * (1 if model_parallel else world) * S / dt

    # Parity of the run itself ("make the 8-GPU tier check, not just time"): a digest of the state the timed steps ended in --
    # model ids, poses, confidence thresholds, surfel counts (summed over their owners), the label mask.  With object models spread over
    # the GPUs, rank 0 then plays the SAME frames through a single-GPU instance and compares: the line says whether N GPUs gave the bits
    # one GPU gives.  (Independent replicas: rank 0's sequence is the N = 1 sequence; its digest is the same at every N.)
    parity = None
    try:
        parity = parity_vs_n1(args, torch, dist, facade, cf, streams[0], rank, world, local_rank, model_parallel, W, H, cam, P, pre_masks, use_gt, gt_mask)
    except Exception as e:  # noqa: BLE001 -- the headline line must not depend on this leg
        parity = dict(error=str(e))
    replicas = None
    if world > 1 and not args.no_extras:
        try:
            replicas = replicas_leg(args, torch, facade, local_rank, rank, world, barrier, all_reduce_max)
        except Exception as e:  # noqa: BLE001 -- the headline line must not depend on this leg
            replicas = dict(error=str(e))
    out = None
    if rank == 0:
        n_models = cf.num_models
        counts = [cf.model_info(i)["count"] for i in range(n_models)]
        achieved = (prof.icp_bytes / 1e9) / (prof.icp_ms_total / 1e3) if prof.icp_ms_total > 0 else 0.0
        bpl = int(prof.icp_bytes / max(1, prof.icp_launches))
        avg_us = 1e3 * prof.icp_ms_total / max(1, prof.icp_launches)
        # PHYSICAL bytes of the launch: what the kernel visits, from what the kernel reports -- the frame's vertex + normal
        # planes once (24 B per pixel, shared by the trackers through L2), per tracker 24 B for every pixel of the 64-pixel runs inside its
        # final screen box (the whole image for the background) and 11 B for every pixel of the record slots between its first and last RGB
        # candidate (cf_odom_level0_visited; the last timed frame's tracking call)
        visited = [cf.model_level0_visited(i) for i in range(n_models)]
        # (the residual pass reads the 1-byte candidate mask over the whole slot range and the other 10 bytes -- next depth, both intensities,
        # gathered last depth -- only where the mask is set; a culled tracker's mask is empty outside its prediction, i.e. outside the pixels
        # its ICP runs cover: 10 B x min(range, ICP pixels).  A tracker that is not culled: 11 B x every pixel, SURVEY 8(d)'s figure.)
        def residual_bytes(icp_px, res_px):
            return 11 * res_px if res_px >= W * H else res_px + 10 * min(res_px, icp_px)
        processed = 24 * W * H + sum(24 * v[0] + residual_bytes(v[0], v[1]) for v in visited)
        # (the committed counter passes and rocprofv3 durations are those of the one-GPU launch: with the trackers spread over ranks this
        # rank's launch is another one, and neither is quoted)
        traffic, traffic_source = pmc_traffic(args.workload, W * H) if world == 1 else (None, "counter passes exist for the one-GPU launch only; at "
                                                                                        "N > 1 this rank's launch carries its own trackers only")
        if world > 1:
            pmc_traffic.rocprof_us = None
        phys_gbs = processed / max(avg_us, 1e-9) / 1e3
        roofline = dict(bound="hbm", achieved=round(phys_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(phys_gbs / HBM_PEAK_GBS, 4),
                        traffic=traffic, traffic_source=traffic_source,
                        kernel="cf::icp_reduce_kernel<PPT,%d>: ICP reduction of all lock-step models || their RGB residual passes, pyramid level 0"
                               % (4 if n_models > 1 else 0),
                        launches=int(prof.icp_launches), avg_us=round(avg_us, 3), bytes_per_launch=int(processed),
                        sampled="level-0 launches of every %d-th timed step carry begin/end events" % args.event_sampling,
                        bytes_per_pixel="24 (frame planes, once) + per tracker 24 x ICP pixels visited + residual pass: 11 x every pixel (unculled) or 1 x slot-range pixels + 10 x min(slot range, ICP pixels) (culled: the candidate mask is empty outside the prediction)",
                        pixels_visited=dict(icp=[int(v[0]) for v in visited], residual=[int(v[1]) for v in visited], frame=W * H),
                        frac_reference_work=round(achieved / HBM_PEAK_GBS, 4), reference_work_gbs=round(achieved, 1), reference_work_bytes_per_launch=bpl,
                        convention="`achieved` / `frac` = PHYSICAL bytes per launch / the launch's own duration: the pixels the kernel visits (a culled tracker "
                                   "only walks the 64-pixel runs inside its screen box and the record slots between its first and last RGB candidate; both come "
                                   "from the tracker's own state after the last timed frame) -- what `traffic` (PMC counters of the same launch) should and does "
                                   "agree with.  `frac_reference_work` is SURVEY 8(d) read literally, (24 + 24 M + 11 M) B x every pixel of the frame for M "
                                   "trackers: the reference's work done per second, which culling makes larger than any memory system could deliver (it was this "
                                   "line's `frac` until round 5)")
        if pmc_traffic.rocprof_us:   # the same launches in the committed rocprofv3 --kernel-trace run of this build (shorter: see the file)
            roofline.update(rocprofv3_avg_us=pmc_traffic.rocprof_us, frac_rocprofv3=round(processed / pmc_traffic.rocprof_us / 1e3 / HBM_PEAK_GBS, 4),
                            rocprofv3_source="profiles/r5*_icp_level0_timed_launches.txt: kernel durations of the timed steps' level-0 launches under rocprofv3 "
                                             "--kernel-trace (every dispatch followed by an idle gap); `avg_us` above is this process's own begin / end events on the plain "
                                             "stream, back to back with the launch in front of it -- the conservative figure, and the one `frac` is quoted on")
        # the surfel stage against ITS roofline: SURVEY 8(d)'s 384 B per surfel and model-frame over the stream time of the
        # stage's chain of batched launches (index maps, association, compactions, update, clean, prediction), sampled like the ICP launch
        surf = None
        if prof.surfel_calls:
            s_us = 1e3 * prof.surfel_ms_total / prof.surfel_calls
            s_gbs = (prof.surfel_bytes / 1e9) / (prof.surfel_ms_total / 1e3)
            surf = dict(bound="hbm", achieved=round(s_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(s_gbs / HBM_PEAK_GBS, 4),
                        stage="cf_models_frame_passes: 2 index maps + association + update + clean + 2 compactions + prediction of all models, 10 batched launches (the update beside the first compaction's block sums, the second index pass's rasterisation beside that compaction)",
                        chains=int(prof.surfel_calls), avg_us=round(s_us, 1), bytes_per_chain=int(prof.surfel_bytes / prof.surfel_calls), surfels=int(sum(counts)),
                        bytes_per_surfel="8 passes x 48 B for a fusing model (SURVEY 8d: 2 index + 2 splat reads, update R+W, clean R+W)",
                        note="latency-bound: ~330 k surfels are 0.13 GB per frame -- 16 us at the HBM peak -- behind ten dependent launches with scatter / "
                             "gather passes (atomicMin z-keys, 4x4 window gathers); the image-space outputs (56 B per pixel and index map, 38 B per pixel "
                             "of the prediction) are not in the byte count")
        out = dict(metric="frames/sec at 640x480 (N active models) + ICP-reduce achieved HBM GB/s vs peak", value=round(fps, 2),
                   unit="frames/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 4),
                   higher_is_better=True, scaling="strong" if model_parallel else "weak", vs_baseline=None, dtype="f32", data="synthetic",
                   input="resident",
                   config=dict(workload=f"{wl['config']}: {wl['desc']}, {W}x{H} synthetic noisy RGB-D, whole CoFusion::processFrame hot path "
                                        "(bilateral, tracking SO3+4/5/10 ICP+RGB GN, segmentation, predict, fuse, clean)",
                               active_models=n_models, object_models=n_models - 1, surfels=counts,
                               preroll_frames=P, preroll_masks=pre_masks or ("gt" if use_gt else ("crf" if n_obj else "none")),
                               segmentation=("none (-static)" if n_obj == 0 else
                                             "ground-truth masks" if use_gt else
                                             "SLIC + exact O(K^2) dense-CRF mean field as specified by oracle/orc_segment.c (gSLICr / densecrf are not in "
                                             "the reference tree: parity of this stage is against the oracle only)"),
                               icp_launch=[args.icp_threads, args.icp_ppt], icp_arith=args.icp_arith, gn_mode=args.gn_mode,
                               streams_per_gpu=S, parallel=args.parallel if world > 1 else "single",
                               rccl_world=(world if use_rccl else 0),
                               placement=(placement_of([cf.model_info(i)["id"] for i in range(n_models)], world, n_obj >= world) if model_parallel else "one GPU"),
                               collectives=("library RCCL communicator (ncclBroadcast + ncclAllReduce in place on the context's stream)" if use_rccl
                                            else "torch.distributed callbacks" if model_parallel else "none"),
                               metric_definition="r04: configs[2] up to 5 GPUs (its five models), configs[3] beyond (8 objects, one per GPU, the background "
                                                 "on rank 0); pre-roll with ground-truth masks, warm-up and timed steps with the motion CRF; `replicas` = N "
                                                 "independent configs[2] sequences",
                               background="split over the ranks (replicated map; surfel-range index map + row-band ICP with all-reduce)" if (model_parallel and args.shard_background) else "one rank",
                               frames=("broadcast from rank 0 every step, consumed in stream order" if model_parallel else "ring of device-resident frames, complete before each call (device_frames_complete=1)")),
                   roofline=roofline)
        if surf is not None:
            out["roofline_surfel"] = surf
        if parity is not None:
            out["parity_vs_n1"] = parity
        if replicas is not None:
            out["replicas"] = replicas
        if world == 1 and S == 1 and not args.no_extras:
            extras(out, args, cf, cam, frames, base + args.steps, use_gt, torch, facade, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            try:
                c1, call = cpu_baseline(cf, cam, frames, base + args.steps + 200, step_stream, streams[0], args.cpu_budget)
                out["cpu_baseline"] = c1
                out["cpu_baseline_all_cores"] = call
            except Exception as e:  # noqa: BLE001 -- the headline line must not depend on this leg
                out["cpu_baseline"] = dict(error=str(e))
        print(json.dumps(out), file=out_stream, flush=True)
    for st in streams:
        st["cf"].close()
    if dist is not None:
        dist.destroy_process_group()
    return out


def state_digest(cf, counts=None):
    """sha256 over (number of models; per model: id, pose bits, confidence threshold bits, surfel count) + the label mask"""
    import hashlib
    h = hashlib.sha256()
    n = cf.num_models
    h.update(np.int32(n).tobytes())
    for i in range(n):
        info = cf.model_info(i)
        h.update(np.int32(info["id"]).tobytes()); h.update(np.ascontiguousarray(info["pose"], np.float32).tobytes())
        h.update(np.float32(info["conf_threshold"]).tobytes())
        h.update(np.int64(info["count"] if counts is None else counts[i]).tobytes())
    h.update(np.ascontiguousarray(cf.mask()).tobytes())
    return h.hexdigest()


def parity_vs_n1(args, torch, dist, facade, cf, st, rank, world, local_rank, model_parallel, W, H, cam, P, pre_masks, use_gt, gt_mask):
    total = P + args.warmup + args.steps
    if not model_parallel:
        return dict(sha256=state_digest(cf), frames_played=total,
                    what="ids, poses, confidence thresholds, surfel counts, label mask after the timed steps; the same at every N (rank 0's sequence)") if rank == 0 else None
    n = cf.num_models
    dev = torch.device("cuda", local_rank)
    own = torch.tensor([cf.model_info(i)["count"] if cf.model_owned(i) else 0 for i in range(n)], dtype=torch.int64,
                       device=dev if dist.get_backend() == "nccl" else "cpu")
    owners = torch.tensor([1 if cf.model_owned(i) else 0 for i in range(n)], dtype=torch.int64, device=own.device)
    dist.all_reduce(own); dist.all_reduce(owners)
    if args.shard_background and n:   # every rank holds a replica of the split background: counted once
        own[0] = own[0] // max(1, int(owners[0].item()))
    out = None
    if rank == 0:
      try:   # (whatever happens here, rank 0 meets the others at the barrier below)
        sha_par = state_digest(cf, counts=[int(v) for v in own.cpu().tolist()])
        single = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels, enable_multiple_models=1,
                                 device_frames_complete=1)
        if args.icp_ppt:
            single.set_icp_launch(args.icp_threads, args.icp_ppt)
        if args.gn_mode >= 0:
            single.set_gn_mode(args.gn_mode)
        single.set_icp_arith(args.icp_arith)   # (explicitly: the same rounding specification as the ranks', whatever the environment says)
        for i in range(total):
            k = frame_index(i, args.frames)
            f = st["frames"][k]
            if (i < P and pre_masks == "gt") or use_gt:
                single.process_frame(f["depth"], f["rgb"], mask=gt_mask(f), timestamp=i)
            else:
                single.process_frame_device(st["resident"][k]["depth"], st["resident"][k]["rgba"], timestamp=i)
        sha_one = state_digest(single)
        single.close()
        out = dict(identical=(sha_par == sha_one), sha256=sha_par, sha256_one_gpu=sha_one, frames_played=total,
                   what="rank 0 replayed the same frames through a single-GPU instance after the timed steps: ids, poses, confidence thresholds, "
                        "surfel counts (from their owners), label mask")
      except Exception as e:  # noqa: BLE001
        out = dict(error=str(e))
    dist.barrier()
    return out


def placement_of(model_ids, world, colocate):
    """model id -> rank, as host/CoFusion.h Distributed::owner places them (co_fusion_amd/parallel.assign_models restates it)"""
    from co_fusion_amd import parallel
    pl = parallel.assign_models(list(model_ids), world, colocate=colocate)
    return {str(m): r for r, ms in pl.items() for m in ms}


def replicas_leg(args, torch, facade, local_rank, rank, world, barrier, all_reduce_max):
    """N > 1, second figure of the line: one INDEPENDENT configs[2] sequence per GPU (own seed, no data-path collective), the same
    pre-roll / warm-up / K timed steps / barriers / MAX over ranks as the headline -> aggregate frames/s over the N replicas
    ("weak": per-GPU work fixed).  The strong-scaling headline of a 1.5 ms latency-bound frame cannot grow with N; this one shows
    what N GPUs deliver on N streams."""
    wl = WORKLOADS["objects4"]
    W, H = wl["size"]
    dev = torch.device("cuda", local_rank)
    cam, frames = make_stream(W, H, args.frames, n_obj=wl["n_obj"], seed=1234 + 64 * rank)
    cfi = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=1 << 21, enable_multiple_models=1,
                          device_frames_complete=1)
    res = [dict(depth=torch.from_numpy(f["depth"]).to(dev), rgba=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
    P = 24 * wl["n_obj"] + 30
    for i in range(P):
        f = frames[frame_index(i, args.frames)]
        cfi.process_frame(f["depth"], f["rgb"], mask=(f["label"] * 40).astype(np.uint8), timestamp=i)

    def step(i):
        k = frame_index(i, args.frames)
        cfi.process_frame_device(res[k]["depth"], res[k]["rgba"], timestamp=i)

    dt = timed_region(lambda i: step(P + i), args.steps, args.warmup, barrier, all_reduce_max)
    n_models = cfi.num_models
    cfi.close()
    return dict(value=round(world * args.steps / dt, 2), unit="frames/s (aggregate over N independent sequences)", scaling="weak",
                ms_per_step=round(1e3 * dt / args.steps, 4), workload="configs[2] per GPU, one independent sequence each, no collective",
                active_models_rank0=n_models)


def lockstep_run(args, torch, facade, local_rank, wl, S):
    """--streams S --lockstep [--groups G]: S independent sequences (different seeds) in G lock-step groups on one GPU (G = 1: ONE group, one
    context; G > 1: each group has its own context, HIP stream and host thread).  A step = one frame of EVERY sequence; value =
    S * steps / time (aggregate frames/s).  Same phases as the single-sequence run."""
    import threading
    W, H = args.width, args.height
    n_obj = wl["n_obj"]
    dev = torch.device("cuda", local_rank)
    G = max(1, min(args.groups, S))
    seqs = [make_stream(W, H, args.frames, n_obj=n_obj, seed=1234 + si) for si in range(S)]
    cam = seqs[0][0]
    members = [list(range(g, S, G)) for g in range(G)]
    groups = []
    for g in range(G):
        hs = torch.cuda.Stream(device=dev) if G > 1 else None
        if hs is not None:
            torch.cuda.set_stream(hs)   # the group adopts the current stream at construction
        grp = facade.CoFusionGroup(len(members[g]), W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels,
                                   enable_multiple_models=int(n_obj > 0), device_frames_complete=1)
        if args.icp_ppt:
            grp.sequences[0].set_icp_launch(args.icp_threads, args.icp_ppt)   # (the launch shape belongs to the shared context)
        if args.gn_mode >= 0:
            grp.sequences[0].set_gn_mode(args.gn_mode)
        groups.append(dict(g=grp, stream=hs, members=members[g]))
    torch.cuda.set_stream(torch.cuda.default_stream(dev))
    res = [[dict(depth=torch.from_numpy(f["depth"]).to(dev), rgba=torch.from_numpy(f["rgba"]).to(dev)) for f in fr] for _, fr in seqs]
    torch.cuda.synchronize()

    def step(gr, i, masks=None):
        k = frame_index(i, args.frames)
        if masks == "gt":
            fs = [seqs[m][1][k] for m in gr["members"]]
            gr["g"].process_frames([f["depth"] for f in fs], [f["rgb"] for f in fs], [(f["label"] * 40).astype(np.uint8) for f in fs], timestamp=i)
        else:
            gr["g"].process_frames_device([res[m][k]["depth"] for m in gr["members"]], [res[m][k]["rgba"] for m in gr["members"]], timestamp=i)

    def run(lo, hi, masks=None):
        if G == 1:
            for i in range(lo, hi):
                step(groups[0], i, masks)
            return
        def work(gr):
            torch.cuda.set_device(local_rank)
            for i in range(lo, hi):
                step(gr, i, masks)
        ths = [threading.Thread(target=work, args=(gr,)) for gr in groups]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    P = args.preroll
    pre_masks = "gt" if (n_obj > 0 and args.preroll_masks == "gt") else None
    run(0, P, pre_masks)
    run(P, P + args.warmup)
    base = P + args.warmup
    cf0 = groups[0]["g"].sequences[0]
    cf0.profile_enable(0 if args.no_kernel_events else args.event_sampling)
    cf0.profile_read(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(base, base + args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = cf0.profile_read(reset=True)
    cf0.profile_enable(False)
    n_models = [q.num_models for gr in groups for q in gr["g"].sequences]
    trackers = sum(q.num_models for q in groups[0]["g"].sequences)
    avg_us = 1e3 * prof.icp_ms_total / max(1, prof.icp_launches)
    achieved = (prof.icp_bytes / 1e9) / (prof.icp_ms_total / 1e3) if prof.icp_ms_total > 0 else 0.0
    out = dict(metric="frames/sec at 640x480 (N active models) + ICP-reduce achieved HBM GB/s vs peak", value=round(S * args.steps / dt, 2),
               unit="frames/s (aggregate over the sequences)", n_gpus=1, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 4),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic", input="resident",
               config=dict(workload=f"{wl['config']}: {wl['desc']}, {W}x{H} synthetic noisy RGB-D -- {S} independent sequences in {G} lock-step group(s) "
                                    "(per group: one context, one set of tracking launches for the trackers of its sequences)",
                           sequences=S, groups=G, active_models=n_models, trackers_per_launch=min(trackers, 16), preroll_frames=P, preroll_masks=pre_masks or "none",
                           streams_per_gpu=S, parallel="lock-step group" if G == 1 else "lock-step groups on their own streams / host threads"),
               roofline=dict(bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                             traffic=None, kernel="cf::icp_reduce_kernel<PPT,4>: level-0 launch with the trackers of all sequences of the first group",
                             launches=int(prof.icp_launches), avg_us=round(avg_us, 3), bytes_per_launch=int(prof.icp_bytes / max(1, prof.icp_launches)),
                             bytes_per_pixel="counted as 24 + (24 + 11) * trackers, i.e. the current-frame planes ONCE although the sequences do not share them (an undercount)"))
    print(json.dumps(out))
    for gr in groups:
        gr["g"].close()
    return out


def dry_run(args, rank, world, dist, out_stream=None):
    """No GPU: stub steps through the same timing contract and process-group plumbing (CPU test of `--gpus N`)."""
    import torch

    def barrier():
        if dist is not None:
            dist.barrier()

    def all_reduce_max(dt):
        if dist is None:
            return dt
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    dt = timed_region(lambda i: time.sleep(0.001 * (1 + rank)), args.steps, args.warmup, barrier, all_reduce_max)
    out = None
    if rank == 0:
        out = dict(metric="dry run (no GPU work)", value=round(args.steps / dt, 2), unit="frames/s", n_gpus=world, steps=args.steps,
                   warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 4), higher_is_better=True,
                   scaling="strong" if args.parallel == "models" else "weak", vs_baseline=None, dtype="f32", data="none",
                   config=dict(workload=f"dry run of {args.workload} ({WORKLOADS[args.workload]['config']})", parallel=args.parallel if world > 1 else "single",
                               placement=placement_of(range(WORKLOADS[args.workload]["n_obj"] + 1), world, WORKLOADS[args.workload]["n_obj"] >= world)))
        print(json.dumps(out), file=out_stream or sys.stdout, flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return out


def extras(out, args, cf, cam, frames, i0, use_gt, torch, facade, local_rank):
    """Untimed-for-the-headline extras on rank 0: host-input rate (+ trajectory error over those frames), oracle trajectory check,
    and the static configuration as `secondary`."""
    W, H = args.width, args.height
    n = len(frames)
    try:
        K = max(20, min(args.steps, 60))
        torch.cuda.synchronize()
        est, gt = [], []
        t0 = time.perf_counter()
        for i in range(i0, i0 + K):
            f = frames[frame_index(i, n)]
            cf.process_frame(f["depth"], f["rgb"], mask=(f["label"] * 40).astype(np.uint8) if use_gt else None, timestamp=i)
            est.append(cf.model_info(0)["pose"][:3, 3].astype(np.float64))  # host copy of the pose: no device access
            gt.append(f["T"][:3, 3])
        torch.cuda.synchronize()
        dth = time.perf_counter() - t0
        err = np.linalg.norm(np.array(est) - np.array(gt), axis=1)
        out["host_input"] = dict(value=round(K / dth, 2), unit="frames/s", ms_per_step=round(1e3 * dth / K, 4), steps=K,
                                 note="same stream through cofusion_process_frame (CoFusion.cpp:179-184 semantics): pageable host depth f32 + "
                                      "rgb u8x3 -> pinned staging -> async H2D, RGB->RGBA on the device; PCIe inclusive")
        out["ate_m"] = dict(vs_synthetic_gt=round(float(np.sqrt(np.mean(err ** 2))), 6), frames=K,
                            after_frames=i0, note="RMSE of the camera position against the generator's trajectory (no alignment: both "
                                                  "start at the identity); depth noise + %d played frames of drift" % (i0 + K))
    except Exception as e:  # noqa: BLE001
        out["host_input"] = dict(error=str(e))
    try:
        out["klg_input"] = klg_input_leg(args, cam, frames, torch, facade, local_rank)
    except Exception as e:  # noqa: BLE001
        out["klg_input"] = dict(error=str(e)[:300])
    try:
        out["ate_m"] = dict(out.get("ate_m", {}), **oracle_trajectory_check(cam, frames, torch, facade, local_rank, args))
    except Exception as e:  # noqa: BLE001
        out.setdefault("ate_m", {})["vs_oracle_error"] = str(e)
    if args.workload != "static":
        out["secondary"] = secondary_static(args, torch, facade, local_rank)


def klg_input_leg(args, cam, frames, torch, facade, local_rank, n_frames=60):
    """the number to compare the day car4-noise.klg is available.  The workload's synthetic stream is written as a
    .klg log in the format of the reference's recordings (GUI/Tools/KlgLogReader.cpp:22-87: int32 frame count, per frame int64
    timestamp, int32 sizes, zlib-compressed uint16 millimetre depth, JPEG colour) and played through the library's own reader
    (host/KlgIO.cpp + host/Jpeg.cpp: inflate + baseline JPEG decode on the host) into the same processFrame as the headline: frames/s
    INCLUDING log decoding and the host-input upload, and the decode cost alone."""
    import io
    import struct
    import tempfile
    import zlib
    from co_fusion_amd import klg
    W, H = args.width, args.height
    n_obj = WORKLOADS[args.workload]["n_obj"]
    F = n_frames   # (the generated stream is short and played back and forth, frame_index)
    try:
        from PIL import Image
        def colour(rgb):
            buf = io.BytesIO(); Image.fromarray(rgb).save(buf, format="JPEG", quality=90, subsampling=2); return buf.getvalue(), "JPEG (quality 90, 4:2:0)"
    except ImportError:   # (Pillow is in this image; raw colour is what the reader's other branch takes, KlgLogReader.cpp:72-75)
        def colour(rgb):
            return np.ascontiguousarray(rgb).tobytes(), "raw RGB (Pillow not importable)"
    path = os.path.join(tempfile.gettempdir(), f"bench_{os.getpid()}.klg")
    fmt = ""
    nbytes = 4
    with open(path, "wb") as f:
        f.write(struct.pack("<i", F))
        for t in range(F):
            fr = frames[frame_index(t, len(frames))]
            mm = np.rint(fr["depth"] * np.float32(1000.0)).astype(np.uint16)
            zd = zlib.compress(mm.tobytes(), 6)
            jb, fmt = colour(fr["rgb"])
            f.write(struct.pack("<qii", t * 33333, len(zd), len(jb))); f.write(zd); f.write(jb)
            nbytes += 16 + len(zd) + len(jb)
    try:
        t0 = time.perf_counter()
        for _ in klg.KlgReader(path, W, H):
            pass
        decode_ms = 1e3 * (time.perf_counter() - t0) / F
        g = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels, enable_multiple_models=int(n_obj > 0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ts, depth, rgb in klg.KlgReader(path, W, H):
            g.process_frame(depth, rgb, timestamp=ts)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_models = g.num_models
        g.close()
    finally:
        os.remove(path)
    return dict(value=round(F / dt, 2), unit="frames/s", ms_per_step=round(1e3 * dt / F, 4), frames=F, decode_ms_per_frame=round(decode_ms, 4),
                log_bytes_per_frame=int(nbytes / F), depth="zlib(uint16 mm)", colour=fmt, active_models_at_end=n_models,
                note="synthetic stream of this workload written in the reference's .klg format and read back by the library's reader (one host thread: "
                     "inflate + baseline JPEG decode, then the host-input upload) into processFrame from frame 0 (bootstrap, models spawn on the way) -- "
                     "decoding is serial with the GPU work here, the reference's LogReader does the same on its one thread")


def oracle_trajectory_check(cam, frames, torch, facade, local_rank, args, n_frames=8):
    """Free run of the first frames of the HEADLINE workload's pipeline on the GPU and on the CPU oracle (multi-model frame loop with the
    motion CRF for the object workloads, -static otherwise; fast spawning so that object models exist within the checked frames): the
    trajectories of the camera and of every model must be identical (ATE 0, same bits)."""
    W, H = args.width, args.height
    multi = WORKLOADS[args.workload]["n_obj"] > 0
    import orc
    orc.set_icp_arith(args.icp_arith)
    if multi:
        import orc_multi as om
        ref = om.MultiPipeline(cam, conf_global=0.5, spawn_offset=2)
        g = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels, enable_multiple_models=1,
                            conf_global_init=0.5, model_spawn_offset=2)
    else:
        import orc_pipeline as op
        ref = op.StaticPipeline(cam)
        g = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels, enable_multiple_models=0)
    errs, same, most = [], True, 1
    for t in range(n_frames):
        f = frames[t]
        if multi:
            ref.process_frame(f["depth"], f["rgba"])
            rposes = [m.pose for m in ref.models]; rcounts = [m.surfels.shape[0] for m in ref.models]
        else:
            rp, rn = ref.process_frame(f["depth"], f["rgba"])
            rposes, rcounts = [rp], [rn]
        g.process_frame(f["depth"], f["rgb"], timestamp=t)
        same = same and g.num_models == len(rposes)
        most = max(most, len(rposes))
        for i in range(min(g.num_models, len(rposes))):
            info = g.model_info(i)
            same = same and bool((info["pose"].view(np.uint32) == np.asarray(rposes[i], np.float32).view(np.uint32)).all()) and info["count"] == rcounts[i]
        errs.append(float(np.linalg.norm(g.model_info(0)["pose"][:3, 3].astype(np.float64) - np.asarray(rposes[0])[:3, 3].astype(np.float64))))
    g.close()
    return dict(vs_oracle=round(float(np.sqrt(np.mean(np.square(errs)))), 9), vs_oracle_frames=n_frames, vs_oracle_bit_identical=same,
                vs_oracle_pipeline=("multi-model frame loop, motion CRF (the headline workload's pipeline), up to %d models" % most) if multi else "-static",
                vs_reference=reference_trajectory_check(args, multi))


def reference_trajectory_check(args, multi):
    """north_star's parity clause against the REFERENCE'S OWN tracker, measured live, under both arithmetics (see the two parts below): the HIP facade plays a scenario of
    tests/golden/ref_traj_v1.npz (the pinned frame loop tracked by the reference's own RGBDOdometry class under the CPU emulator: f32 tree
    reductions, Eigen-style solve) -- the headline pipeline's two-object motion-CRF scenario at 640x480 for object workloads, the static
    one otherwise -- and tests/trajpin.compare returns the figures: camera ATE, frames with identical model lists, surfel-count
    differences, every object the reference keeps for >= 10 frames.  Test infrastructure (tests/), untimed."""
    try:
        import trajpin
        name = "crf_two_objects_640" if multi else "static_camera_640"
        # (1) THE PARITY CLAUSE: the same stream under the reference-order arithmetic (cf_set_icp_arith 2: the reference's own f32 trees at
        # GPUConfig.h's launch shapes, its host loop) -- model lists, surfel counts and poses must EQUAL the reference tracker's
        ex = None
        try:
            pe, ie, ce = trajpin.play_facade(name, "reference")
            ex = trajpin.exact(name, pe, ie, ce)
            if multi and "gt_masks_two_boxes_640" in trajpin.scenarios():
                p3, i3, c3 = trajpin.play_facade("gt_masks_two_boxes_640", "reference")
                ex["well_conditioned_objects"] = trajpin.exact("gt_masks_two_boxes_640", p3, i3, c3)
        except Exception as e:  # noqa: BLE001
            ex = dict(error=str(e)[:300])
        if args.icp_arith == "reference":
            return dict(reference_order=ex, scenario=name, source="live: HIP facade under cf_set_icp_arith 2 against tests/golden/ref_traj_v1.npz")
        # (2) the DEFAULT arithmetic of the timed path (exact integer sums: launch-shape independent, not the reference's rounding): how far
        # that moves a trajectory -- camera ATE asserted against BASELINE.json's 1e-3 m, counts and objects reported
        poses, ids, counts = trajpin.play_facade(name, args.icp_arith)
        rep = trajpin.compare(name, poses, ids, counts, arith=args.icp_arith, log=lambda s: None)
        tight = None
        if multi and "gt_masks_two_boxes_640" in trajpin.scenarios():
            p2, i2, c2 = trajpin.play_facade("gt_masks_two_boxes_640", args.icp_arith)
            r2 = trajpin.compare("gt_masks_two_boxes_640", p2, i2, c2, arith=args.icp_arith, log=lambda s: None)
            tight = dict(scenario="gt_masks_two_boxes_640", frames=r2["frames"], camera_rmse=round(r2["rmse"], 9), camera_max=round(r2["max"], 9),
                         lists_identical_frames=r2["lists_identical_frames"],
                         objects={k: dict(frames=v["frames"], max_m=round(v["max_m"], 7), within_tight_bound=v["within_tight_bound"], stable_in_reference=v["stable_in_reference"],
                                          count_max_rel_diff=round(v["count_max_rel_diff"], 5)) for k, v in r2["objects"].items()})
        return dict(reference_order=ex, arith=args.icp_arith,
                    rmse=round(rep["rmse"], 9), max=round(rep["max"], 9), frames=rep["frames"], scenario=name, well_conditioned_objects=tight,
                    lists_identical_frames=rep["lists_identical_frames"], count_first_diff_frame=rep["count_first_diff_frame"],
                    count_max_abs_diff=rep["count_max_abs_diff"], count_max_rel_diff=round(rep["count_max_rel_diff"], 7),
                    background_count_max_abs_diff=rep["background_count_max_abs_diff"],
                    background_count_max_rel_diff=round(rep["background_count_max_rel_diff"], 7),
                    objects={k: dict(frames=v["frames"], max_m=round(v["max_m"], 7), within_tight_bound=v["within_tight_bound"]) for k, v in rep["objects"].items()},
                    bound_m=1e-3, source="live: HIP facade on the scenario's stream against tests/golden/ref_traj_v1.npz (frame loop tracked by the "
                                         "reference's own RGBDOdometry class, oracle/ref_shim); `reference_order` = the same under cf_set_icp_arith 2 "
                                         "(identical: lists, surfel counts and poses equal the reference tracker's on every frame), the figures beside it = "
                                         "the default exact-integer arithmetic of the timed path")
    except AssertionError as e:
        return dict(error="bound violated: " + str(e)[:300])
    except Exception as e:  # noqa: BLE001
        return dict(error=str(e)[:300])


def secondary_static(args, torch, facade, local_rank, warmup=30, steps=120):
    """configs[1]: single static background model, same timing rules (frames resident, sync on both sides)."""
    try:
        W, H = args.width, args.height
        cam, frames = make_stream(W, H, args.frames, n_obj=0, seed=1234)
        cf = facade.CoFusion(W, H, cam.fx, cam.fy, cam.cx, cam.cy, device=local_rank, max_surfels=args.max_surfels, enable_multiple_models=0,
                             device_frames_complete=1)
        if args.icp_ppt:
            cf.set_icp_launch(args.icp_threads, args.icp_ppt)
        if args.gn_mode >= 0:
            cf.set_gn_mode(args.gn_mode)
        dev = torch.device("cuda", local_rank)
        res = [dict(depth=torch.from_numpy(f["depth"]).to(dev), rgba=torch.from_numpy(f["rgba"]).to(dev)) for f in frames]
        for i in range(warmup):
            k = frame_index(i, args.frames)
            cf.process_frame_device(res[k]["depth"], res[k]["rgba"], timestamp=i)
        cf.profile_enable(args.event_sampling)
        cf.profile_read(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            k = frame_index(i, args.frames)
            cf.process_frame_device(res[k]["depth"], res[k]["rgba"], timestamp=i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = cf.profile_read(reset=True)
        counts = [cf.model_info(0)["count"]]
        cf.close()
        ach = (prof.icp_bytes / 1e9) / (prof.icp_ms_total / 1e3) if prof.icp_ms_total > 0 else 0.0
        traffic, traffic_source = pmc_traffic("static", W * H)
        return dict(workload="configs[1]: single static background model (-static), synthetic", value=round(steps / dt, 2),
                    unit="frames/s", ms_per_step=round(1e3 * dt / steps, 4), warmup=warmup, steps=steps, active_models=1, surfels=counts,
                    roofline=dict(bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                                  avg_us=round(1e3 * prof.icp_ms_total / max(1, prof.icp_launches), 3),
                                  bytes_per_launch=int(prof.icp_bytes / max(1, prof.icp_launches)), kernel="cf::icp_reduce_kernel<PPT,0>",
                                  traffic=traffic, traffic_source=traffic_source,
                                  note="one unculled tracker: SURVEY 8(d)'s formula, the pixels visited and the counter traffic are the same quantity here "
                                       "((24 + 24 + 11) B x every pixel) -- the launch in which nothing hides behind culling"))
    except Exception as e:  # the headline line must not depend on this extra
        return dict(error=str(e))


def kernel_source_sha():
    """sha256 of the file that holds the dominant kernel: a committed counter pass is only quoted for the build it was taken on"""
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "co_fusion_amd", "csrc", "track_reduce.hip"), "rb").read()).hexdigest()


def pmc_traffic(workload, pixels):
    """(HBM-side bytes per launch of the level-0 ICP kernel, where it comes from) from the committed rocprofv3 PMC passes -- FETCH_SIZE and
    WRITE_SIZE cannot be collected from inside this process.  The newest pass is used, and only if it was taken on THIS source of the
    kernel (kernel_source_sha256 in the JSON, tools/make_traffic_json.py): a stale figure is reported as null, with the reason."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_icp_traffic.json")), reverse=True)
    sha = kernel_source_sha()
    for path in files:
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        for e in (t if isinstance(t, list) else [t]):
            if e.get("workload") == workload and e.get("pixels") == pixels:
                name = os.path.basename(path)
                if e.get("kernel_source_sha256") != sha:
                    return None, (f"profiles/{name} was taken on another build of csrc/track_reduce.hip (sha256 {str(e.get('kernel_source_sha256'))[:12]} != "
                                  f"{sha[:12]}): not quoted; re-run tools/gpu_pmc.sh + tools/make_traffic_json.py")
                pmc_traffic.rocprof_us = e.get("rocprofv3_avg_us")
                return int(e["traffic_bytes_per_launch"]), (f"HBM-side bytes per launch from the committed rocprofv3 --pmc passes of this workload on this build (profiles/{name}: "
                                                            "FETCH_SIZE x2 + WRITE_SIZE x1, factors measured by tools/microbench/fetch_calib.hip); counters cannot "
                                                            "be read from inside the benchmark process")
        break   # only the newest pass counts
    return None, "no committed counter pass for this workload"


pmc_traffic.rocprof_us = None


def native_oracle():
    """The oracle compiled for THIS box's cores (-O3 -march=native -fopenmp); -ffp-contract=off is kept, so its
    results are the oracle's bits."""
    path = os.path.join(ROOT, "oracle", "_build", "liborc_native.so")
    stamp = path + ".host"
    host = open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0] if os.path.exists("/proc/cpuinfo") else "?"
    if not os.path.exists(path) or not os.path.exists(stamp) or open(stamp).read() != host:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"])
        open(stamp, "w").write(host)
    return path


def cpu_baseline(cf, cam, frames, i0, step_stream, st, budget_s):
    """CPU oracle odometry path (port of the reference's tracking: RGBDOdometry map preparation + getIncrementalTransformation for every
    active model), timed on this box's host cores on frames sampled from the same run."""
    import ctypes
    os.environ["ORC_LIB"] = native_oracle()
    import orc
    import orc_pipeline as op
    orc.set_icp_arith({"gram": "gram", "1": "gram", "reference": "reference", "2": "reference"}.get(os.environ.get("CF_ICP_ARITH", ""), "product"))   # (the rounding specification the GPU legs ran with)
    n = len(frames)
    ncpu = orc.usable_cpus()   # affinity mask and cgroup quota, not the number of CPUs the box shows
    omp = ctypes.CDLL("libgomp.so.1")
    # sample: a few consecutive frames of the running sequence; inputs are what the GPU tracker of each model is about to read
    samples = []
    for s in range(3):
        i = i0 + s
        k = frame_index(i, n)
        prev = frames[frame_index(i - 1, n)]
        models = []
        for m in range(cf.num_models):
            v4, n4, img = cf.model_tracking_inputs(m)
            models.append(dict(v4=v4, n4=n4, img=img, pose=cf.model_info(m)["pose"].copy()))
        f = frames[k]
        omp.omp_set_num_threads(ncpu)
        samples.append(dict(models=models, rgba=f["rgba"], prev_rgba=prev["rgba"], pyr=orc.depth_pyramid(op.bilateral(f["depth"], 5.0))))
        step_stream(st, i)
    od = orc.Odometry(cam.width, cam.height, cam.cx, cam.cy, cam.fx, cam.fy)

    def one_frame(smp):
        t = 0.0
        for md in smp["models"]:
            od.init_first_rgb(smp["prev_rgba"])  # SO3 reference image = the previous frame (untimed: state of the previous call)
            t0 = time.perf_counter()
            od.init_icp_model(md["v4"], md["n4"], md["pose"])
            od.init_rgb_model(md["img"])
            od.init_icp(smp["pyr"], 20.0)
            od.init_rgb(smp["rgba"])
            od.track(md["pose"][:3, 3], md["pose"][:3, :3], icp_weight=10.0, so3=True)
            t += time.perf_counter() - t0
        return t

    def leg(threads, max_frames):
        omp.omp_set_num_threads(threads)
        one_frame(samples[0])  # warm-up
        times = []
        t_start = time.perf_counter()
        while len(times) < max_frames and (time.perf_counter() - t_start < budget_s or len(times) < 3):
            times.append(one_frame(samples[len(times) % len(samples)]))
        med = float(np.median(times))
        return dict(value=round(1.0 / med, 3), unit="frames/s", ms_per_frame=round(1e3 * med, 2), cores=threads, kind="port",
                    sample=f"median of {len(times)} frame trackings ({sum(times):.1f} s) over {len(samples)} frames sampled from this run x "
                           f"{len(samples[0]['models'])} active models each; oracle odometry path (map preparation + SO3 + 4/5/10 ICP+RGB GN "
                           f"iterations per model), gcc -O3 -march=native -fopenmp, {threads} of {ncpu} usable host threads ({os.cpu_count()} visible)")

    return leg(1, 100), leg(ncpu, 100)


if __name__ == "__main__":
    main()
