"""Synthetic RGB-D frames and the benchmark configurations.

Host-side data generation only (torch CPU): the reference reads real datasets through
datasets/dataset.py:232-316 and yields ``{frame_id, c2w, rgb, depth, direction}``;
there is no network here, so frames are rendered analytically: a pinhole camera inside
an axis-aligned box room, depth = z-depth of the ray/box hit, colour = a smooth function
of the hit point, a seeded 2 % of pixels with depth 0 (exercises the ``d <= 0`` branch of
model/scene_rep.py:160).
"""
from __future__ import annotations

import copy
import math
from typing import Dict

import numpy as np
import torch


def camera_rays(H: int, W: int, fx: float, fy: float, cx: float, cy: float) -> torch.Tensor:
    """Per-pixel view directions in camera coordinates, OpenGL convention
    (datasets/utils.py:4-41): [(i-cx)/fx, -(j-cy)/fy, -1], shape [H,W,3]."""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def look_rotation(yaw: float, pitch: float) -> torch.Tensor:
    cy_, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    Ry = torch.tensor([[cy_, 0.0, sy], [0.0, 1.0, 0.0], [-sy, 0.0, cy_]])
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]])
    return Ry @ Rx


def render_box_frame(bound, c2w: torch.Tensor, H, W, fx, fy, cx, cy, shrink=0.3, drop=0.02, seed=0,
                     frame_id=0) -> Dict[str, torch.Tensor]:
    """One frame of the box room.  ``bound`` [3,2]; the room is ``bound`` shrunk by ``shrink``."""
    bound = torch.as_tensor(bound, dtype=torch.float32)
    lo, hi = bound[:, 0] + shrink, bound[:, 1] - shrink
    dirs = camera_rays(H, W, fx, fy, cx, cy)
    R, t = c2w[:3, :3].float(), c2w[:3, 3].float()
    d_world = torch.sum(dirs[..., None, :] * R, -1)               # R @ d
    wall = torch.where(d_world > 0, hi, lo)
    tt = (wall - t) / torch.where(d_world.abs() < 1e-9, torch.full_like(d_world, 1e-9), d_world)
    tt = torch.where(d_world.abs() < 1e-9, torch.full_like(tt, float("inf")), tt)
    depth = tt.min(-1).values                                      # z-depth because d_cam.z == -1
    hit = t + d_world * depth[..., None]
    rgb = 0.5 + 0.5 * torch.sin(4.0 * hit)
    g = torch.Generator().manual_seed(seed)
    dead = torch.rand(H, W, generator=g) < drop
    depth = torch.where(dead, torch.zeros_like(depth), depth)
    return {"frame_id": frame_id, "c2w": c2w.clone(), "rgb": rgb.contiguous(), "depth": depth.contiguous(),
            "direction": dirs}


_BASE = {
    "grid": {"enc": "HashGrid", "tcnn_encoding": True, "hash_size": 19, "voxel_sdf": 0.04,
             "use_bound_normalize": True},
    "pos": {"enc": "Frequency", "n_bins": 8},
    "cam": {"H": 480, "W": 640, "fx": 320.0, "fy": 320.0, "cx": 319.5, "cy": 239.5, "crop_edge": 10,
            "near": 0, "far": 5, "depth_trunc": 100.0},
    "data": {"sc_factor": 1},
    "training": {"rgb_weight": 1.0, "depth_weight": 0.0, "sdf_weight": 1000, "fs_weight": 10,
                 "n_samples_d": 50, "range_d": 0.2, "n_range_d": 25, "n_samples": 75, "perturb": 1,
                 "norm_factor": 1.0, "trunc": 0.1, "rgb_missing": 0.0},
    "mapping": {"bound": [[-0.6, 2.95], [0.5, 7.05], [-1.15, 3.05]], "localMLP_max_len": [7.0, 7.0, 7.0],
                "sample": 1800, "pixels_cur": 800, "iters": 15, "lr_embed": 0.01, "lr_decoder": 0.01,
                "lr_rot": 0.001, "lr_trans": 0.001, "map_every": 3, "map_accum_step": 1,
                "pose_accum_step": 5, "map_wait_step": 0, "first_iters": 500},
    "tracking": {"iter_RO": 5, "iter": 10, "sample": 1000, "lr_rot": 0.001, "lr_trans": 0.001,
                 "RO": {"particle_size": 2000, "n_rows": 16, "n_cols": 24}},
}


def config_reference_defaults() -> dict:
    """FastCaMo-synth apartment_2 as the reference ships it
    (configs/FastCaMo-synth/FastCaMo-synth.yaml, apartment_2.yaml:4): S = 50 + 25."""
    return copy.deepcopy(_BASE)


def config_plumbing() -> dict:
    """BASELINE config 1: 32x32 frame, 256 rays x 16 samples, hash 2^10 (small fixtures)."""
    c = copy.deepcopy(_BASE)
    c["grid"]["hash_size"] = 10
    c["cam"].update(H=32, W=32, fx=32.0, fy=32.0, cx=15.5, cy=15.5, crop_edge=0)
    c["training"].update(n_samples_d=11, n_range_d=5, n_samples=16)
    c["mapping"].update(bound=[[-1.0, 1.0], [-1.0, 1.0], [-1.0, 1.0]], sample=256)
    return c


def config_headline() -> dict:
    """BASELINE config 2: apartment_2, 4096 rays x 64 samples (43 uniform + 21 depth-guided,
    the reference's 2:1 ratio), hash 2^19, 640x480 cropped by 10."""
    c = copy.deepcopy(_BASE)
    c["training"].update(n_samples_d=43, n_range_d=21, n_samples=64)
    c["mapping"].update(sample=3296, pixels_cur=800)
    return c


def config_large_submap() -> dict:
    """BASELINE config 4 per-GPU submap: FastCaMo-large uses hash 2^16 and centre-length
    normalisation (configs/FastCaMo-large/floor1.yaml:21-22, FastCaMo-large.yaml:87)."""
    c = config_headline()
    c["grid"]["hash_size"] = 16
    c["grid"]["use_bound_normalize"] = False
    return c


def config_scannet() -> dict:
    """BASELINE config 5 workload: ScanNet scene0000_00 (configs/ScanNet/scannet.yaml, scene0000.yaml:4,21-22):
    the scene bound, far = 7, S = 50 + 25, pose_accum_step 2, and the intrinsics AFTER the reference's floor-division
    quirk (datasets/dataset.py:29-30: ``cfg["cam"]["fx"] // downsample`` floors 577.59 -> 577.0 etc.).  The reference
    itself sets iter_RO 0 for ScanNet; the RandomOptimizer shape [2000, 16 x 24] is kept for the build-side
    stress of the forward-only slice that BASELINE config 5 names."""
    c = copy.deepcopy(_BASE)
    c["cam"].update(fx=577.590698 // 1, fy=578.729797 // 1, cx=318.905426 // 1, cy=242.683609 // 1, far=7)
    c["mapping"].update(bound=[[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]], localMLP_max_len=[7.0, 7.0, 4.0],
                        pose_accum_step=2, map_every=3, iters=10, sample=2000, pixels_cur=500)
    c["tracking"].update(iter_RO=0, iter=10, sample=1000, ignore_edge_W=20, ignore_edge_H=20)
    c["tracking"]["RO"].update(initial_scaling_factor=0.02, rescaling_factor=0.5)
    return c


def intrinsics_after_crop(cfg: dict):
    cam = cfg["cam"]
    e = cam["crop_edge"]
    return cam["H"] - 2 * e, cam["W"] - 2 * e, cam["fx"], cam["fy"], cam["cx"] - e, cam["cy"] - e


def default_pose(cfg: dict, yaw=0.3, pitch=-0.1) -> torch.Tensor:
    b = np.array(cfg["mapping"]["bound"], dtype=np.float64)
    c2w = torch.eye(4)
    c2w[:3, :3] = look_rotation(yaw, pitch)
    c2w[:3, 3] = torch.tensor(b.mean(1), dtype=torch.float32)
    return c2w


def make_frame(cfg: dict, c2w=None, seed=0, frame_id=0):
    H, W, fx, fy, cx, cy = intrinsics_after_crop(cfg)
    if c2w is None:
        c2w = default_pose(cfg)
    return render_box_frame(cfg["mapping"]["bound"], c2w, H, W, fx, fy, cx, cy, seed=seed, frame_id=frame_id)


def ray_batch(frame: Dict[str, torch.Tensor], indices: torch.Tensor, c2w: torch.Tensor):
    """Gather pixels -> (rays_o, rays_d, target_rgb, target_d) as mipsfusion.py:179-185 does."""
    H, W = frame["depth"].shape
    r, c = torch.div(indices, W, rounding_mode="floor"), torch.remainder(indices, W)
    d_cam = frame["direction"][r, c]
    rays_d = torch.sum(d_cam[..., None, :] * c2w[:3, :3], -1)
    rays_o = c2w[None, :3, 3].repeat(indices.shape[0], 1)
    return rays_o.contiguous(), rays_d.contiguous(), frame["rgb"][r, c].contiguous(), frame["depth"][r, c][:, None].contiguous()
