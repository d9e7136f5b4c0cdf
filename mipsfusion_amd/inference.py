"""Forward-only consumers of the path: full-image rendering and dense grid queries.

``render_full_img`` is ``Logger.render_full_img`` (Logger.py:193-214) without the Logger: the image's rays go through
``model.render_rays`` in chunks and the per-ray colour and depth are stitched back.  ``get_batch_query_fn`` /
``query_in_batches`` are Mesher's batching helper (model/Mesher.py:12-18, 342-344, 392-394, 487-489, 628-630) for
the ``query_*`` entry points that take ALREADY-normalised coordinates.

Both accept a (rank, world) pair: this is the one place where ray data parallelism across the GPUs of a node pays
(SURVEY 8e): the rays / points are split into ``world`` contiguous shares, every rank renders its share with its own
replica of the sub-map, and one ``all_gather`` (RCCL on GPUs) reassembles the image.  No gradient is involved.
"""
from typing import Callable, Optional, Tuple

import torch

from . import dist as mdist


def rays_camera_to_world(rays_d_cam: torch.Tensor, c2w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """geometry_helper.rays_camera_to_world (geometry_helper.py:107-123): -> rays_d [n,3], rays_o [n,3]."""
    rays_d = torch.sum(rays_d_cam[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[None, :3, -1].repeat(rays_d.shape[0], 1)
    return rays_d, rays_o


share_of = mdist.share_of


@torch.no_grad()
def render_full_img(model, rays_d_cam: torch.Tensor, pose_local: torch.Tensor, gt_depth: Optional[torch.Tensor],
                    H: int, W: int, ray_batch_size: int = 10000, rank: int = 0, world: int = 1, noise=None):
    """-> rgb [H,W,3], depth [H,W] (on the model's device).  rays_d_cam [H,W,3] or [H*W,3] camera-frame directions,
    pose_local [4,4], gt_depth [H,W] or None (no depth guidance).  ``noise``: optional [H*W,S] jitter table
    (extension, as in JointEncoding.forward) -- by default the reference's CPU torch.rand draw per chunk."""
    dev = model.embed_fn.params.device
    d_cam = torch.reshape(rays_d_cam, (-1, 3)).to(dev, torch.float32)
    depth = None if gt_depth is None else torch.reshape(gt_depth, (-1, 1)).to(dev, torch.float32)
    rays_d, rays_o = rays_camera_to_world(d_cam, pose_local.to(dev, torch.float32))
    n = rays_d.shape[0]
    begin, end = share_of(n, rank, world)
    rgb_list, depth_list = [], []
    for i in range(begin, end, ray_batch_size):
        j = min(i + ray_batch_size, end)
        out = model.render_rays(rays_o[i:j].contiguous(), rays_d[i:j].contiguous(),
                                None if depth is None else depth[i:j].contiguous(),
                                noise=None if noise is None else noise[i:j])
        rgb_list.append(out["rgb"])
        depth_list.append(out["depth"])
    rgb = torch.cat(rgb_list, 0) if rgb_list else torch.empty((0, 3), device=dev)
    dep = torch.cat(depth_list, 0) if depth_list else torch.empty((0,), device=dev)
    if world > 1:
        rgb = mdist.all_gather_ragged(rgb, n, world)
        dep = mdist.all_gather_ragged(dep, n, world)
    return rgb.reshape(H, W, 3), dep.reshape(H, W)


def get_batch_query_fn(query_fn: Callable, args_num: int = 1) -> Callable:
    """model/Mesher.py:12-18."""
    if args_num == 1:
        return lambda f, i0, i1: query_fn(f[i0:i1, ...])
    return lambda f, i0, i1, v: query_fn(f[i0:i1, ...], v)


@torch.no_grad()
def query_in_batches(query_fn: Callable, pts_normalised: torch.Tensor, batch_size: int = 1024 * 16, rank: int = 0,
                     world: int = 1) -> torch.Tensor:
    """Mesher's evaluation loop over a dense grid of ALREADY-normalised points [n,3] -> [n,C]."""
    fn = get_batch_query_fn(query_fn)
    n = pts_normalised.shape[0]
    begin, end = share_of(n, rank, world)
    outs = [fn(pts_normalised, i, min(i + batch_size, end)) for i in range(begin, end, batch_size)]
    out = torch.cat(outs, 0) if outs else None
    if world > 1:
        out = mdist.all_gather_ragged(out, n, world)
    return out
