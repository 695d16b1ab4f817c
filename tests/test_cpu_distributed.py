"""CPU suite: the N > 1 code paths under torch.distributed with the gloo backend, world_size 2."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import common
        import orc
        from co_fusion_amd import parallel

        # 1. bench.py timing contract: barrier on both sides, MAX over ranks, whole-job aggregate
        def step(i):
            time.sleep(0.002 * (rank + 1))      # rank 1 is the slow one
        dt = bench.timed_region(step, steps=10, warmup=2, barrier=dist.barrier,
                                all_reduce_max=lambda s: parallel.allreduce_max_seconds(s))
        assert dt >= 10 * 0.002 * world * 0.9, dt     # everybody reports the slowest rank's time

        # 2. image-row sharded ICP: per-rank partial fixed-point sums, one int64 SUM all-reduce == full-image sums
        W, H = 160, 120
        fp = common.frame_pair(W, H)
        cam = fp["cam"]; ocam = orc.Cam(cam.fx, cam.fy, cam.cx, cam.cy)
        pose = common.perturbed_pose(2)
        od = orc.Odometry(W, H, cam.cx, cam.cy, cam.fx, cam.fy)
        od.init_first_rgb(fp["rgba0"]); od.init_icp_model(fp["v4"], fp["n4"], pose); od.init_rgb_model(fp["img"])
        od.init_icp(orc.depth_pyramid(fp["d1"]), 20.0); od.init_rgb(fp["rgba1"])
        vc, nc, vp, npv = (od.buffer(k, 0) for k in range(4))
        Rinv = np.linalg.inv(pose[:3, :3].astype(np.float64)).astype(np.float32)
        angle = np.float32(np.sin(20.0 * 3.14159254 / 180.0))
        full, _ = orc.icp_step(pose[:3, :3], pose[:3, 3], vc, nc, Rinv, pose[:3, 3], ocam, vp, npv, 0.10, angle)
        band = parallel.row_bands(H, world)[rank]
        vc_band = vc.copy()
        keep = np.zeros(H, bool); keep[band.start:band.stop] = True
        vc_band[:H][~keep] = np.nan                  # pixels of other ranks: NaN x plane == "not mine"
        part, _ = orc.icp_step(pose[:3, :3], pose[:3, 3], vc_band, nc, Rinv, pose[:3, 3], ocam, vp, npv, 0.10, angle)
        t = torch.from_numpy(part.copy())
        parallel.allreduce_se3_sums(t)
        assert np.array_equal(t.numpy(), full), "sharded + all-reduced sums must equal the single-device sums exactly"
        assert part[28] < full[28]

        # 3. surfel-range sharded z-buffer: MIN all-reduce of u64 keys held in int64 tensors (unsigned order!)
        rng = np.random.default_rng(5)
        allkeys = rng.integers(0, 1 << 63, size=(world, 64), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(world, 64), dtype=np.uint64)
        allkeys[0, :8] = np.uint64(0xFFFFFFFFFFFFFFFF)      # empty pixels on rank 0
        allkeys[:, 8:12] = np.uint64(0xFFFFFFFFFFFFFFFF)    # empty everywhere
        mine = torch.from_numpy(allkeys[rank].view(np.int64).copy())
        parallel.allreduce_min_keys(mine)
        assert np.array_equal(mine.numpy().view(np.uint64), allkeys.min(axis=0)), "u64 MIN through the signed collective"

        # 4. placement
        pl = parallel.assign_models([0, 3, 5, 9], world)
        assert sorted(sum(pl.values(), [])) == [0, 3, 5, 9] and 0 in pl[0]
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs: p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_assign_models_and_bands_single_process():
    from co_fusion_amd import parallel
    assert parallel.assign_models([0, 1, 2], 1) == {0: [0, 1, 2]}
    pl = parallel.assign_models([0, 1, 2, 3, 4, 5, 6, 7, 8], 8)
    assert pl[0] == [0] and all(len(v) >= 1 for v in pl.values())
    # BASELINE.json configs[3]: 8 object models one per GPU, the background sharing rank 0 (cofusion_config.colocate_background)
    pc = parallel.assign_models([0, 1, 2, 3, 4, 5, 6, 7, 8], 8, colocate=True)
    assert pc[0] == [0, 1] and all(pc[r] == [r + 1] for r in range(1, 8))
    bands = parallel.row_bands(480, 8)
    assert bands[0].start == 0 and bands[-1].stop == 480 and sum(len(b) for b in bands) == 480


def test_bench_gpus_2_respawns_and_reports_two_ranks():
    """`python bench.py --gpus 2` from a plain shell re-executes itself under torch.distributed.run (one process per rank),
    runs the timing contract over the process group and prints ONE JSON line with n_gpus == 2 (round 1 silently ran one rank).
    --dry-run replaces processFrame by a stub step: there is no GPU here."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2
    assert out["scaling"] == "strong" and out["config"]["parallel"] == "models"   # default partition of N > 1: models over ranks
    assert out["ms_per_step"] >= 2.0 * 0.9        # MAX over ranks: rank 1 sleeps 2 ms per step


def test_bench_gpus_8_defaults_to_configs3_with_one_object_per_rank():
    """the driver's `--gpus 8` line runs BASELINE.json's own 8-GPU configuration (configs[3]: 8 objects + background, one object model
    per GPU) and says so; up to 5 GPUs the metric's configs[2].  --dry-run: no GPU here."""
    import json
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse(["--gpus", "1"]).workload == "objects4" and bench.parse(["--gpus", "4"]).workload == "objects4"
    assert bench.parse(["--gpus", "8"]).workload == "objects8" and bench.parse(["--gpus", "8", "--workload", "objects4"]).workload == "objects4"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 8 and "configs[3]" in out["config"]["workload"]
    pl = out["config"]["placement"]
    assert len(pl) == 9 and pl["0"] == 0 and sorted(pl[str(k)] for k in range(1, 9)) == list(range(8)), pl
