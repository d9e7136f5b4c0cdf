"""Multi-GPU sharding of the hot path: one process per GPU, one (or ceil(n/world)) submap per process.

Submaps are independent optimisation problems (own grid, decoder, Adam state, keyframe rays;
InactiveMap.py:28,66-70,207-210); the only coupling is the small table of keyframe poses.  So there is no
data-path collective: after a BA round each rank publishes the poses it optimised with one all_gather
(K x 7 floats, latency bound).  Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.
"""
from typing import List

import torch
import torch.distributed as dist


def submaps_of_rank(n_submaps: int, world: int, rank: int) -> List[int]:
    """Round-robin ownership (the reference's inactive-map loop is round-robin too, InactiveMap.py:207-210)."""
    return [s for s in range(n_submaps) if s % world == rank]


def exchange_poses(rot: torch.Tensor, trans: torch.Tensor, group=None) -> torch.Tensor:
    """rot [K,4], trans [K,3] of THIS rank's submap -> [world, K, 7] (quaternion | translation) on every rank.
    A no-op (returns [1,K,7]) when torch.distributed is not initialised."""
    mine = torch.cat([rot.detach(), trans.detach()], -1).contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        return mine[None]
    world = dist.get_world_size(group)
    dev = mine.device
    if dist.get_backend(group) == "gloo" and mine.is_cuda:      # debugging the N>1 path without RCCL
        mine = mine.cpu()
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return torch.stack(out, 0).to(dev)


def max_over_ranks(seconds: float, device, group=None) -> float:
    """Timing convention of bench.py: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    if dist.get_backend(group) == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t)


def all_gather_ragged(mine: torch.Tensor, n_total: int, world: int, group=None) -> torch.Tensor:
    """Contiguous shares of an [n_total, ...] tensor (share sizes differ by at most one, see
    inference.share_of) -> the whole tensor on every rank.  One all_gather of equal-size padded blocks."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return mine
    base, extra = divmod(n_total, world)
    block = base + (1 if extra else 0)
    dev = mine.device
    buf = mine
    if dist.get_backend(group) == "gloo" and buf.is_cuda:
        buf = buf.cpu()
    pad = torch.zeros((block,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    pad[:buf.shape[0]] = buf
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    parts = [out[r][:base + (1 if r < extra else 0)] for r in range(world)]
    return torch.cat(parts, 0).to(dev)
