"""Cross-sub-map global bundle adjustment, sharded one sub-map per GPU (BASELINE config 4).

Reference: ``InactiveMap.global_BA_overlapping`` (InactiveMap.py:375-474) with ``get_SDF_dif`` / ``get_SDF_dif2`` /
``infer_pts`` (InactiveMap.py:128-192) and ``compute_avg_SDF_difference`` (helper_functions/geometry_helper.py:225-229).
The unknowns are the world poses of the sub-maps' first keyframes (all but sub-map 0's, (n-1) x 7 numbers); every loss
term couples two adjacent sub-maps i, j: surface points seen by keyframes both sub-maps contain are carried into each
sub-map's local frame through its anchor pose and the two networks must predict the same SDF there.

The reference evaluates both networks of every pair in one process.  Here each rank owns the networks of ITS
sub-maps (``mipsfusion_amd.dist.submaps_of_rank``) and the anchor poses are replicated:

  1. every rank evaluates the SDF predictions of the sides it owns (autograd reaches the replicated anchor parameters
     through the points: the hash grid's d feat / d x, ``mipsf_hashgrid_dx_from_jac``);
  2. one SUM all-reduce of the [terms, 2, batch] prediction table gives every rank the other sides' values
     (a few tens of KB);
  3. every rank forms the loss with its own sides differentiable and the foreign sides constant and back-propagates:
     d L / d anchors splits into the path through side i (on i's owner) and the path through side j (on j's owner);
  4. one SUM all-reduce of the (n-1) x 7 pose gradients per pose step -- the collective SURVEY 8e names -- and the
     identical Adam step on every rank.

With one rank (or a rank that owns both sides of a pair) the same code is the reference's single-process loop.
The network query is injected (``query(submap_id, pts_local[N,3]) -> sdf[N]`` in network units) so that the
collective logic is testable on CPU tensors with gloo.
"""
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch

from . import dist as mdist
from .helper_functions.geometry_helper import matrix_to_quaternion, qt_to_transform_matrix


class PairTerm:
    """One loss term: sub-maps (i, j), rays [N,7] = (camera-frame direction | rgb | depth), the world pose [N,4,4] (or
    [1,4,4]) of the keyframe each ray belongs to, an optional validity mask [N,1] (default: depth > 0, as
    ``get_SDF_dif``; ``get_SDF_dif2`` passes the overlap mask), a weight (5 per pair, 100 for the loop-closing
    keyframe: InactiveMap.py:428, 447)."""

    def __init__(self, i: int, j: int, rays: torch.Tensor, kf_pose_world: torch.Tensor, weight: float = 5.0,
                 mask: Optional[torch.Tensor] = None):
        self.i, self.j, self.rays, self.kf_pose_world, self.weight = int(i), int(j), rays, kf_pose_world, float(weight)
        d = rays[..., 6:7]
        self.mask = mask if mask is not None else torch.where(d > 0., torch.ones_like(d), torch.zeros_like(d))


def points_in_submap(anchor_inv: torch.Tensor, kf_pose_world: torch.Tensor, rays: torch.Tensor) -> torch.Tensor:
    """``infer_pts`` up to the network call (InactiveMap.py:128-134): back-projected surface points of the rays in the
    sub-map's local frame.  anchor_inv [4,4] = inverse of the sub-map's first-keyframe world pose."""
    local = anchor_inv @ kf_pose_world                                  # [N,4,4] (or [1,4,4])
    d_cam, depth = rays[..., :3], rays[..., 6:7]
    rays_d = torch.sum(d_cam[..., None, None, :] * local[..., None, :3, :3], -1)
    rays_o = local[..., None, :3, -1].repeat(1, rays_d.shape[1], 1).reshape(-1, 3)
    if rays_o.shape[0] == 1:
        rays_o = rays_o.expand(rays_d.shape[0], 3)
    rays_d = rays_d.reshape(-1, 3)
    return (rays_o[..., None, :] + rays_d[..., None, :] * depth[..., :, None]).reshape(-1, 3)


def avg_sdf_difference(a: torch.Tensor, b: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """geometry_helper.compute_avg_SDF_difference (geometry_helper.py:225-229)."""
    return torch.sum(torch.square(a * mask - b * mask)) / (torch.count_nonzero(mask) + 0.001)


class ShardedGlobalBA:
    def __init__(self, query: Callable[[int, torch.Tensor], torch.Tensor], owned: Iterable[int],
                 first_kf_poses: torch.Tensor, trunc: float, lr_rot: float = 1e-3, lr_trans: float = 1e-3,
                 pose_accum_step: int = 1, group=None, adam=None):
        """first_kf_poses [n,4,4]: world poses of the sub-maps' first keyframes (row 0 stays fixed,
        InactiveMap.py:399); owned: the sub-map ids whose networks this rank evaluates -- ownership must be DISJOINT
        across ranks (``dist.submaps_of_rank``): a side evaluated twice would be summed twice."""
        self.query, self.owned, self.trunc, self.group = query, set(int(s) for s in owned), float(trunc), group
        self.accum, self.it = int(pose_accum_step), 0
        p = first_kf_poses.detach()
        self.fixed = p[:1].clone()
        self.cur_trans = torch.nn.Parameter(p[1:, :3, 3].clone())
        self.cur_rot = torch.nn.Parameter(matrix_to_quaternion(p[1:, :3, :3]))
        adam = adam or torch.optim.Adam
        self.opt = adam([{"params": self.cur_rot, "lr": lr_rot}, {"params": self.cur_trans, "lr": lr_trans}])
        self.opt.zero_grad()

    def anchors(self) -> torch.Tensor:
        return torch.cat([self.fixed, qt_to_transform_matrix(self.cur_rot, self.cur_trans)], 0)

    def _predict(self, sid: int, anchor_inv, term: PairTerm) -> torch.Tensor:
        pts = points_in_submap(anchor_inv, term.kf_pose_world, term.rays)
        return self.query(sid, pts).reshape(-1, 1) * self.trunc          # pred_sdf (InactiveMap.py:138)

    def iteration(self, terms: Sequence[PairTerm]) -> torch.Tensor:
        """One iteration of the loop at InactiveMap.py:408-459 -> the loss value (identical on every rank)."""
        anchors = self.anchors()
        inv = torch.linalg.inv(anchors)
        dev = anchors.device
        n_max = max(t.rays.shape[0] for t in terms)
        table = torch.zeros((len(terms), 2, n_max), dtype=torch.float32, device=dev)
        own = {}
        for k, t in enumerate(terms):
            for side, sid in ((0, t.i), (1, t.j)):
                if sid in self.owned:
                    s = self._predict(sid, inv[sid], t)
                    own[(k, side)] = s
                    table[k, side, :s.shape[0]] = s.detach().reshape(-1)
        mdist.all_reduce_sum_(table, self.group)
        loss_local, loss_value = None, torch.zeros((), dtype=torch.float32, device=dev)
        for k, t in enumerate(terms):
            n = t.rays.shape[0]
            a = own.get((k, 0), table[k, 0, :n].reshape(-1, 1))
            b = own.get((k, 1), table[k, 1, :n].reshape(-1, 1))
            term_loss = t.weight * avg_sdf_difference(a, b, t.mask)
            loss_value = loss_value + term_loss.detach()
            if (k, 0) in own or (k, 1) in own:
                loss_local = term_loss if loss_local is None else loss_local + term_loss
        if loss_local is not None and loss_local.requires_grad:
            loss_local.backward()
        self.it += 1
        if self.it % self.accum == 0:
            self._step()
        return loss_value

    def _step(self):
        g_rot = self.cur_rot.grad if self.cur_rot.grad is not None else torch.zeros_like(self.cur_rot)
        g_trans = self.cur_trans.grad if self.cur_trans.grad is not None else torch.zeros_like(self.cur_trans)
        flat = torch.cat([g_rot, g_trans], 1).contiguous()                # (n-1) x 7 floats
        mdist.all_reduce_sum_(flat, self.group)                           # THE collective of SURVEY 8e row 1
        self.cur_rot.grad, self.cur_trans.grad = flat[:, :4].contiguous(), flat[:, 4:].contiguous()
        self.opt.step()
        self.opt.zero_grad()

    def result(self) -> torch.Tensor:
        """World poses of all first keyframes after optimisation (InactiveMap.py:466-471)."""
        return self.anchors().detach()


class frozen:
    """Context manager: the sub-map networks take no gradient while the anchors are optimised (the reference computes
    and discards them)."""

    def __init__(self, models):
        self.params = [p for m in models for p in m.parameters()]

    def __enter__(self):
        self.flags = [p.requires_grad for p in self.params]
        for p in self.params:
            p.requires_grad_(False)

    def __exit__(self, *exc):
        for p, f in zip(self.params, self.flags):
            p.requires_grad_(f)


def model_query(models) -> Callable[[int, torch.Tensor], torch.Tensor]:
    """query(sid, pts) over product sub-map models {sid: JointEncoding}: column 3 of run_network (InactiveMap.py:136-138)."""
    def q(sid, pts):
        return models[sid].run_network(pts)[..., 3]
    return q
