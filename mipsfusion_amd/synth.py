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


def render_rooms_frame(room_a, room_b, door, c2w: torch.Tensor, H, W, fx, fy, cx, cy, drop=0.02, seed=0,
                       frame_id=0) -> Dict[str, torch.Tensor]:
    """One frame of TWO box rooms that share the wall z = room_a[2][1] = room_b[2][0], with a door opening
    ``door`` = [[x0, x1], [y0, y1]] in that wall (BASELINE config 3: a trajectory through several rooms).  A ray that
    leaves the camera's room through the door ends on the other room's walls."""
    A, B = torch.as_tensor(room_a, dtype=torch.float32), torch.as_tensor(room_b, dtype=torch.float32)
    dirs = camera_rays(H, W, fx, fy, cx, cy)
    R, t = c2w[:3, :3].float(), c2w[:3, 3].float()
    d_world = torch.sum(dirs[..., None, :] * R, -1)
    safe = torch.where(d_world.abs() < 1e-9, torch.full_like(d_world, 1e-9), d_world)

    def exit_depth(box):
        wall = torch.where(d_world > 0, box[:, 1], box[:, 0])
        tt = (wall - t) / safe
        tt = torch.where(d_world.abs() < 1e-9, torch.full_like(tt, float("inf")), tt)
        return tt.min(-1)
    z_wall = float(A[2, 1])
    in_a = float(t[2]) < z_wall
    own, other = (A, B) if in_a else (B, A)
    dep, face = exit_depth(own)
    hit = t + d_world * dep[..., None]
    through = (face == 2) & ((d_world[..., 2] > 0) == in_a) & ((hit[..., 2] - z_wall).abs() < 1e-3) & \
        (hit[..., 0] > door[0][0]) & (hit[..., 0] < door[0][1]) & (hit[..., 1] > door[1][0]) & (hit[..., 1] < door[1][1])
    dep2, _ = exit_depth(other)
    depth = torch.where(through, dep2, dep)
    hit = t + d_world * depth[..., None]
    rgb = 0.5 + 0.5 * torch.sin(4.0 * hit)
    g = torch.Generator().manual_seed(seed)
    dead = torch.rand(H, W, generator=g) < drop
    depth = torch.where(dead, torch.zeros_like(depth), depth)
    return {"frame_id": frame_id, "c2w": c2w.clone(), "rgb": rgb.contiguous(), "depth": depth.contiguous(),
            "direction": dirs}


def config_two_rooms() -> dict:
    """BASELINE config 3 workload: the reference's FastCaMo-synth settings as shipped (S = 50 + 25, 1800 + 800 rays, 5 RO
    rounds, 10 tracking / 15 mapping iterations, 500 initialisation iterations per sub-map, 15 pose-only iterations after a
    switch back: FastCaMo-synth.yaml:16-33, 53-55, 73-80) over a bound that covers two rooms joined by a door."""
    c = copy.deepcopy(_BASE)
    c["mapping"]["bound"] = [[-0.6, 2.95], [0.5, 7.05], [-1.15, 7.05]]
    c["mapping"]["keyframe_every"] = 15
    c["tracking"]["switch"] = {"lr_rot": 0.001, "lr_trans": 0.001, "map_num": 15}
    c["cam"]["far"] = 6
    return c


TWO_ROOMS = {"room_a": [[-0.3, 2.65], [0.8, 6.75], [-0.85, 2.75]], "room_b": [[-0.3, 2.65], [0.8, 6.75], [2.75, 6.75]],
             "door": [[0.55, 1.85], [0.8, 5.2]]}


def two_room_sequence(cfg: dict, n_frames: int = 300, kf_every: int = 15):
    """-> (ground-truth poses, frames, schedule): a camera walks a loop in room A, through the door into room B, a loop
    there, and backwards through the door into room A again (4 cm and 0.5 degrees per frame on average, peaks 7.5 cm / 2
    degrees; the camera keeps a corner in view: facing a single textureless-geometry wall of a box room leaves a translation
    along the wall to the weak colour term -- measured: a 20 cm slide in the middle of a half turn, kept by the map from then on).  schedule {frame: ("new",) | ("back", submap)}: a NEW
    sub-map at the first keyframe after the door, back to sub-map 0 at the first keyframe after the return -- the decisions
    Manager.process_keyframe (Manager.py, host control plane, out of scope) would take, fixed in advance."""
    H, W, fx, fy, cx, cy = intrinsics_after_crop(cfg)

    def smooth(a):
        a = min(1.0, max(0.0, a))
        return a * a * (3 - 2 * a)
    # way points in the horizontal (x, z) plane, height y = 3.8; yaw pi looks along +z
    seg = [0.27, 0.20, 0.20, 0.20, 0.13]                    # shares of the sequence: loop A, to B, loop B, back, loop A
    edges = np.cumsum([0.0] + seg)
    poses, room_of = [], []
    for k in range(n_frames):
        u = k / max(1, n_frames - 1)
        s = int(np.searchsorted(edges, u, side="right") - 1)
        s = min(s, len(seg) - 1)
        a = (u - edges[s]) / seg[s]
        if s == 0:                                          # loop in A around (1.2, 0.9), looking outwards then towards the door
            ang = 2 * math.pi * smooth(a)
            x, z, yaw = 1.2 + 0.45 * math.sin(ang), 0.9 - 0.45 * math.cos(ang) + 0.45, math.pi + 0.35 * math.sin(ang)
        elif s == 1:                                        # through the door
            x, z, yaw = 1.2, 0.9 + (3.9 - 0.9) * smooth(a), math.pi
        elif s == 2:                                        # loop in B around (1.2, 4.4)
            ang = 2 * math.pi * smooth(a)
            x, z, yaw = 1.2 + 0.45 * math.sin(ang), 4.35 - 0.45 * math.cos(ang), math.pi - 0.35 * math.sin(ang)
        elif s == 3:                                        # back through the door, walking backwards (still looking along +z)
            x, z, yaw = 1.2, 3.9 + (1.1 - 3.9) * smooth(a), math.pi
        else:
            ang = 2 * math.pi * smooth(a)
            x, z, yaw = 1.2 - 0.2 * math.sin(ang), 1.1 + 0.2 * math.cos(ang) - 0.2, math.pi + 0.25 * math.sin(ang)
        c2w = torch.eye(4)
        c2w[:3, :3] = look_rotation(yaw, -0.08 + 0.04 * math.sin(9.0 * u))
        c2w[:3, 3] = torch.tensor([x, 3.8 + 0.08 * math.sin(5.0 * u), z])
        poses.append(c2w)
        room_of.append(0 if z < TWO_ROOMS["room_a"][2][1] else 1)
    frames = [render_rooms_frame(TWO_ROOMS["room_a"], TWO_ROOMS["room_b"], TWO_ROOMS["door"], poses[k], H, W, fx, fy, cx, cy,
                                 seed=k, frame_id=k) for k in range(n_frames)]
    schedule, active = {}, 0
    for k in range(kf_every, n_frames, kf_every):
        if room_of[k] == 1 and active == 0:
            schedule[k], active = ("new",), 1
        elif room_of[k] == 0 and active == 1:
            schedule[k], active = ("back", 0), 0
    return poses, frames, schedule


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
