"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The reference is single-GPU (no NCCL/MPI/streams anywhere), so this is new design:

* Independent units shard with NO data-path collective: RGB-D streams (bench.py --gpus N) and, inside one
  stream, object models (`assign_models`): each model tracks/predicts/fuses/cleans using only the shared frame
  and its own surfels (Core/CoFusion.cpp:214-217, 465-488).
* Where one model is split (image-row bands of the ICP/RGB reductions, surfel-range shards of the background
  for very large maps) the only exchange is the 6x6 normal-equation system.  The kernels accumulate it as
  exact 64-bit fixed-point integers, so `allreduce_se3_sums` (one [M x 32] int64 SUM all-reduce per
  Gauss-Newton iteration for all M models of the launch) reproduces the single-GPU sums bit for bit,
  independent of the number of ranks and of the reduction order inside RCCL.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def assign_models(model_ids: Sequence[int], world: int, colocate: bool = False) -> Dict[int, List[int]]:
    """Placement of active models on ranks (host/CoFusion.h: Distributed::owner): the background (id 0, by far the largest map)
    alone on rank 0 and object id k on rank 1 + (k - 1) % (world - 1); with `colocate` (cofusion_config.colocate_background,
    BASELINE.json configs[3]: one object model per GPU) object id k on rank (k - 1) % world, the background sharing rank 0."""
    out: Dict[int, List[int]] = {r: [] for r in range(world)}
    for m in model_ids:
        if world <= 1 or m == 0:
            out[0].append(m)
        elif colocate:
            out[(m - 1) % world].append(m)
        else:
            out[1 + (m - 1) % (world - 1)].append(m)
    return out


def row_bands(rows: int, world: int, align: int = 4) -> List[range]:
    """Contiguous image-row bands per rank (aligned so that pyramid levels split at the same place)."""
    per = ((rows + world - 1) // world + align - 1) // align * align
    return [range(min(r * per, rows), min((r + 1) * per, rows)) for r in range(world)]


def allreduce_se3_sums(sums: torch.Tensor) -> torch.Tensor:
    """SUM all-reduce of the fixed-point normal equations, int64 [M, 32] (or [32]).  Exact: integer addition
    commutes, so every rank obtains the bits a single GPU would have produced."""
    assert sums.dtype == torch.int64
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return sums


def allreduce_max_seconds(seconds: float, device=None) -> float:
    """bench.py timing contract: MAX over ranks of the timed region."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sharded_icp_sums(ctx, Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres,
                     rank: int = None, world: int = None) -> torch.Tensor:
    """ONE model's ICP reduction split over the ranks by image-row bands: this rank reduces its band on its GPU
    (cf_icp_step_band), the int64[32] fixed-point sums are SUM-all-reduced (RCCL on GPUs, gloo in tests).  The result is
    bit-identical on every rank to a single-GPU cf_icp_step of the whole image."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    rows = vmap_curr.shape[0] // 3
    band = row_bands(rows, world)[rank]
    if len(band):
        sums = ctx.icp_step_band(Rcurr, tcurr, vmap_curr, nmap_curr, Rprev_inv, tprev, cam, vmap_g_prev, nmap_g_prev, dist_thres, angle_thres,
                                 band.start, band.stop)
    else:
        sums = torch.zeros(32, dtype=torch.int64).numpy()
    t = torch.from_numpy(sums.copy())
    if dist.is_initialized() and dist.get_backend() == "nccl":
        t = t.cuda()
    return allreduce_se3_sums(t).cpu()


_SIGN = -(1 << 63)


def allreduce_min_keys(keys: torch.Tensor) -> torch.Tensor:
    """MIN all-reduce of z-buffer key maps (u64 bit patterns held in int64 tensors).  The keys are ordered as UNSIGNED
    integers; flipping the top bit maps that order onto the signed order the collective compares in."""
    assert keys.dtype == torch.int64
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        k = keys ^ _SIGN
        if dist.get_backend() != "nccl":
            k = k.cpu()
        dist.all_reduce(k, op=dist.ReduceOp.MIN)
        keys.copy_(k.to(keys.device) ^ _SIGN)
    return keys


def sharded_predict_indices(model, pose, time, max_depth, count: int, rank: int = None, world: int = None):
    """predictIndices of ONE surfel map whose surfels are split into contiguous ranges over the ranks: rasterise the own
    range, MIN-all-reduce the key maps, resolve.  Bit-identical to model.predict_indices on one GPU."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    per = (count + world - 1) // world
    keys = model.index_keys(pose, time, max_depth, min(rank * per, count), min((rank + 1) * per, count))
    torch.cuda.synchronize()
    keys = allreduce_min_keys(keys)
    model.index_resolve(pose, keys)
