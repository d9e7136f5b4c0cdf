"""Ray-batch data parallelism for ONE sub-map: the step mipsfusion.py:320-335 shards.

Every rank holds a replica of the sub-map (grid + decoder) and renders its contiguous share of the iteration's rays; the
gradients are dense (the hash table's 36 MB, the decoder's 146 KB, the pose parameters' K x 7 floats), so

    grid      reduce-scatter of the flat gradient  ->  every rank runs Adam on ITS 1/world slice of the table
              (parameters, moments and the 28 B/parameter of Adam traffic divided by world)  ->  all-gather of the updated slices
    decoder   all-reduce of its ten small gradients, the identical (replicated) Adam step on every rank
    poses     all-reduce of the pose gradients, identical step on every rank

-- the bytes of one all-reduce, with the optimiser's work and state divided by the world size.  The objective is the ONE
objective of the whole batch, the step the step of mipsfusion.py:325-335 on the same rays: the loss kernel of every rank
leaves the nine sums its losses are made of (squared errors, front / band / valid-depth counts), one 80-byte all-reduce adds
them (``JointEncoding.ray_share_reduce``), every rank finishes the SAME losses from the batch's sums -- fs_weight / sdf_weight
from the batch's counts (helper_functions/utils.py:43-47), depth_loss over the batch's valid rays (scene_rep.py:218), means
over all N rays -- and differentiates them with respect to its own rays; the ranks' gradients are then SUMMED.  Equal to the
single-process step up to the order of fp32 additions (tests/test_dist_cpu.py on the oracle's modules, the two-process GPU
test on the real JointEncoding).

xGMI is point-to-point: with 8 GPUs a direct schedule moves 7 x 4.5 MB per phase per GPU (~30 us per phase at 153 GB/s per
link), a ring would be per-link bound at ~0.4 ms -- against ~0.1 ms of per-rank compute for 512 rays x 64 samples.  At the
headline batch this sharding is communication-bound (SURVEY 8e says so); it pays for large batches (full-image supervision,
several backward passes accumulated per ``step()``).  UNMEASURED on more than one GPU (no multi-GPU node was
available to this build); covered by a two-rank gloo test on CPU tensors and a two-process run on one GPU.
"""
from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import dist as _mdist
from .dist import all_reduce_sum_, rank_world, share_of


def _padded(n: int, world: int) -> int:
    return (n + world - 1) // world * world


class ShardedFlatAdam:
    """reduce-scatter -> Adam on this rank's slice -> all-gather, for ONE flat fp32 parameter (the hash table).

    param: the replicated flat parameter (a leaf with ``.grad`` accumulated by the caller's backward passes).
    make_optimizer(shard_param) -> an optimiser with ``step()`` over the 1-D slice parameter it is given (FusedAdam on the
    GPU: the slice is a VIEW of ``param``'s storage, so the kernel updates the replica in place; torch.optim.Adam in the CPU
    test).  Gradients arrive as the SUM over ranks (every rank differentiated the whole batch's objective over its rays)."""

    def __init__(self, param: torch.Tensor, make_optimizer: Callable[[torch.nn.Parameter], torch.optim.Optimizer], group=None):
        self.group = group
        self.rank, self.world = rank_world(group)
        self.param = param
        n = param.numel()
        self.n, self.n_pad = n, _padded(n, self.world)
        self.per = self.n_pad // self.world
        self.begin = self.rank * self.per
        self.end = min(n, self.begin + self.per)
        # the slice this rank owns, as a Parameter sharing the replica's storage (padding lives in a private tail)
        flat = param.data.view(-1)
        if self.n_pad == n:
            self._buf = flat
        else:       # padded copy: parameters are exchanged through it (one extra copy per step; the table sizes of the
            self._buf = torch.zeros(self.n_pad, dtype=flat.dtype, device=flat.device)      # reference's configs divide by 8)
            self._buf[:n].copy_(flat)
        self._flat = flat
        self.shard = torch.nn.Parameter(self._buf[self.begin:self.begin + self.per])
        self.shard.grad = torch.zeros_like(self.shard)
        self._grad_pad = None if self.n_pad == n else torch.zeros(self.n_pad, dtype=flat.dtype, device=flat.device)
        self.opt = make_optimizer(self.shard)

    def _backend_is_gloo(self):
        return not self._single() and dist.get_backend(self.group) == "gloo"

    def _single(self):          # (one rank: nothing to exchange -- unless the test switch asks for the collectives anyway)
        return self.world == 1 and not (_mdist.FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())

    @torch.no_grad()
    def step(self):
        """Consumes ``param.grad`` (sum of this rank's backward passes since the last step), leaves it zero."""
        g = self.param.grad.view(-1)
        if self._grad_pad is not None:
            # (padded table: parameters travel through a private copy -- re-read the replica first, it may have been
            # reloaded since the last step: recover_initial_param / load_state_dict)
            self._buf[:self.n].copy_(self.param.data.view(-1))
            self._grad_pad[:self.n].copy_(g)
            g_full = self._grad_pad
        else:
            g_full = g
        out = self.shard.grad
        if self._single():
            out.copy_(g_full[self.begin:self.begin + self.per])
        elif self._backend_is_gloo():
            # gloo has no reduce-scatter: all-reduce and keep the slice (CPU tests; a GPU tensor takes a host round trip)
            h = g_full.detach().cpu() if g_full.is_cuda else g_full.clone()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(h[self.begin:self.begin + self.per])
        else:
            dist.reduce_scatter_tensor(out, g_full, op=dist.ReduceOp.SUM, group=self.group)
        self.opt.step()
        g.zero_()
        if not self._single():
            if self._backend_is_gloo():
                mine = self.shard.data.detach().cpu() if self.shard.is_cuda else self.shard.data.clone()
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                self._buf.copy_(torch.cat(parts).to(self._buf.device))
            else:
                dist.all_gather_into_tensor(self._buf, self.shard.data, group=self.group)
        if self._grad_pad is not None:
            self.param.data.view(-1).copy_(self._buf[:self.n])


class RayDataParallelStep:
    """The optimiser side of a ray-data-parallel mapping iteration over one replicated ``JointEncoding``:

        rdp = RayDataParallelStep(model, make_grid_opt, make_decoder_opt, pose_params, make_pose_opt)
        for it in range(iters):
            rays = my_share(all_rays)                       # ``share_of(N, rank, world)``
            loss = objective(model.forward(*rays)); loss.backward()
            rdp.step(pose=(it + 1) % pose_accum_step == 0)

    The constructor sets ``model.ray_share_reduce`` (``close()`` / leaving the ``with`` block clears it; while it is set every
    training forward is a COLLECTIVE call): the model's training forward then returns the losses of the WHOLE batch
    (one 80-byte all-reduce inside forward), so ``loss.backward()`` leaves every rank with its rays' part of the batch's
    gradient.  ``step`` SUMS the gradients over the ranks (grid: reduce-scatter into this rank's slice; decoder and poses:
    all-reduce), runs the optimisers, and hands every rank the updated table (all-gather).  All ranks end every step with
    bit-identical parameters, equal to the single-process step on the same rays up to fp32 addition order."""

    def __init__(self, model, make_grid_opt, make_decoder_opt, pose_params: Sequence[torch.nn.Parameter] = (),
                 make_pose_opt: Optional[Callable[[List[torch.nn.Parameter]], torch.optim.Optimizer]] = None, group=None):
        # `step` SUMS gradients over the ranks, which is the batch's gradient only when the model's forward finishes the
        # WHOLE batch's losses through this hook; a model without it would silently step along world x the gradient of the
        # mean objective
        if not hasattr(model, "ray_share_reduce"):
            raise TypeError("RayDataParallelStep needs a model whose training forward sums its loss terms over the ranks "
                            "through `model.ray_share_reduce` (mipsfusion_amd.model.JointEncoding does): the gradients of the "
                            "shares are SUMMED, not averaged")
        self.group = group
        self.rank, self.world = rank_world(group)
        self.model = model
        self.grid = ShardedFlatAdam(model.embed_fn.params, make_grid_opt, group)
        self.dec_params = [p for p in model.decoder.parameters()]
        self.dec_opt = make_decoder_opt(self.dec_params)
        self.pose_params = list(pose_params)
        self.pose_opt = make_pose_opt(self.pose_params) if (self.pose_params and make_pose_opt) else None
        self._dec_flat = None
        model.ray_share_reduce = self.reduce_share

    def close(self):
        """Take the hook off the model again: from here on its training forward is an ordinary single-process forward (with the
        hook set EVERY training forward issues a collective -- a forward on a subset of the ranks, rank-0 evaluation in
        train() mode say, would wait for the others forever).  Also the exit of ``with RayDataParallelStep(...) as rdp``."""
        if getattr(self.model, "ray_share_reduce", None) == self.reduce_share:
            self.model.ray_share_reduce = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def reduce_share(self, t: torch.Tensor) -> torch.Tensor:
        """sum of a small fp64 vector over the ranks (the nine loss sums + the ray count of every share)"""
        if self.world > 1 or _mdist.FORCE_COLLECTIVES:
            all_reduce_sum_(t, self.group)
        return t

    def my_share(self, n: int):
        return share_of(n, self.rank, self.world)

    @torch.no_grad()
    def _sum_grads(self, params):
        """one all-reduce for a list of small gradients (flattened into one buffer: a collective per tensor would be ten
        latency-bound calls)"""
        if self.world == 1 and not _mdist.FORCE_COLLECTIVES:
            return
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        all_reduce_sum_(flat, self.group)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    @torch.no_grad()
    def step(self, pose: bool = False):
        if self.model.embed_fn.params.grad is None:
            raise RuntimeError("RayDataParallelStep.step() needs a backward pass first (the table has no gradient)")
        self.grid.step()
        self._sum_grads(self.dec_params)
        self.dec_opt.step()
        torch._foreach_zero_([p.grad for p in self.dec_params if p.grad is not None])
        if pose and self.pose_opt is not None:
            self._sum_grads(self.pose_params)
            self.pose_opt.step()
            torch._foreach_zero_([p.grad for p in self.pose_params if p.grad is not None])
