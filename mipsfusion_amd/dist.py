"""Multi-GPU sharding of the hot path: one process per GPU, one (or ceil(n/world)) submap per process.

Submaps are independent optimisation problems (own grid, decoder, Adam state, keyframe rays;
InactiveMap.py:28,66-70,207-210); the only coupling is the small table of keyframe poses.  So there is no
data-path collective: after a BA round each rank publishes the poses it optimised with one all_gather
(K x 7 floats, latency bound).  Backend: ``nccl`` (= RCCL over xGMI) on GPUs, ``gloo`` in the CPU tests.
"""
from typing import List

import torch
import torch.distributed as dist

# Test switch (tests/test_gpu_dist.py): with one rank the collectives below are skipped -- there is nothing to exchange.  Set, a
# one-rank process group runs them anyway, so that the RCCL entry points (all_reduce, all_gather, reduce_scatter_tensor,
# all_gather_into_tensor) execute on the one GPU a test box has; the results must equal the skipped path's.
FORCE_COLLECTIVES = False


def _single(world: int) -> bool:
    return world == 1 and not FORCE_COLLECTIVES


def share_of(n: int, rank: int, world: int):
    """Contiguous share [begin, end) of n items for `rank` of `world` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def rank_world(group=None):
    """(rank, world) of the default / given process group; (0, 1) when torch.distributed is not initialised."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of a small tensor (pose gradients, prediction tables).  RCCL on GPU tensors; with the
    gloo debugging backend a GPU tensor takes a host round trip.  No-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or _single(dist.get_world_size(group)):
        return t
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
        return t
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


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
    if not (dist.is_available() and dist.is_initialized()) or _single(world):
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


def gather_particle_results(local_rows: torch.Tensor, n_particles: int, group=None) -> torch.Tensor:
    """RandomOptimizer particle split (RandomOptimizer.py:113-131, 196-224): every rank evaluated the
    fitness of its contiguous share of the swarm (`share_of(n_particles, rank, world)`), `local_rows` = [share, C]
    (mean masked |sdf| and the 7-D particle pose per particle); -> [n_particles, C] on every rank, in particle order,
    so that every rank performs the identical swarm update.  One all_gather of n_particles*C floats (64 KB at 2000 x 8)."""
    rank, world = rank_world(group)
    if _single(world):
        return local_rows
    return all_gather_ragged(local_rows.contiguous(), n_particles, world, group=group)
