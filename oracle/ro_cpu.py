"""CPU restatement of the reference's particle-swarm pre-tracker (RandomOptimizer.py:54-227).

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline); never imported by the product.
Pinned: tests/golden/ro.npz holds poses produced by the reference's own class (imported in the build
container with pytorch3d.transforms -> oracle/p3d_cpu, whose quaternion formulas are themselves "parity
unpinned", see p3d_cpu.py) and tests/test_oracle_golden.py checks this file against them.

State carried from one round to the next: rotation [3,3], translation [3,1], search size ([1,6] after the first
round, the scalar initial_scaling_factor before it).
"""
import torch

from . import p3d_cpu

SDF_WEIGHT = 1000.0          # RandomOptimizer.py:23


def pose_6d_to_7d(p6):
    """[qx,qy,qz,tx,ty,tz] -> [qw,...] with qw = sqrt(1 - |q_imag|^2) or 0 (RandomOptimizer.py:57-63)."""
    s = p6[:, 0] ** 2 + p6[:, 1] ** 2 + p6[:, 2] ** 2
    qw = torch.where(s <= 1.0, torch.sqrt(1 - s), torch.zeros_like(s))
    return torch.cat([qw[:, None], p6], -1)


def particle_points(rot, trans, pst7, cam_pts):
    """absolute particle poses applied to the camera-frame points (RandomOptimizer.py:72-88, 121):
    world[p] = (rot @ R(q_p)) @ cam^T + (trans + t_p)."""
    d_r = p3d_cpu.quaternion_to_matrix(pst7[:, :4])
    a_rot = rot @ d_r
    a_trans = trans + pst7[:, 4:, None]
    w = a_rot @ cam_pts.transpose(0, 1) + a_trans
    return w.transpose(1, 2), a_rot, a_trans


def mean_masked_sdf(run_network, world, target_d, trunc):
    """RandomOptimizer.py:117-128: mean over the lattice of valid * |sdf * trunc|."""
    valid = (target_d > 0).to(world.dtype).squeeze(-1)[None]
    sdf = run_network(world)[..., 3] * trunc
    return torch.mean(valid * sdf.abs(), -1)


def swarm_update(mms, pst7, rot, trans, c2):
    """RandomOptimizer.py:196-224: advanced-particle weights, weighted mean transform, pose and search-size update.
    -> rot, trans, search [1,6], info dict."""
    fit = mms * SDF_WEIGHT
    f0 = fit[0]
    better = (fit < f0).to(fit.dtype)
    w = (f0 - fit) * better
    wsum = w.sum() + 0.00001
    ok = bool(torch.count_nonzero(better) > 0)
    if ok:
        mean_sdf = (w * mms).sum() / wsum
        mt = (pst7 * w[:, None]).sum(0) / wsum
        quat = mt[:4] / (mt[:4].norm() + 1e-5)
        mt = torch.cat([quat, mt[4:]])
        rot = rot @ p3d_cpu.quaternion_to_matrix(mt[:4])
        trans = trans + mt[4:, None]
    else:
        mean_sdf = mms[0]
        mt = torch.tensor([1.0, 0, 0, 0, 0, 0, 0], dtype=mms.dtype)
    s = mt[1:].abs() + 0.0001
    search = (c2 * mean_sdf * s / s.norm() + 0.0001)[None]
    if not ok:
        search = search * 2
    return rot, trans, search, {"success": ok, "mean_sdf": mean_sdf, "fitness0": f0, "mean_transform": mt}


@torch.no_grad()
def optimize(run_network, pst, rows, cols, depth_img, rays_dir, initial_pose, n_iter, c1, c2, trunc):
    """-> tracked pose [4,4], per-round trace (list of dicts)."""
    if n_iter <= 0:
        return initial_pose, []
    rot, trans = initial_pose[:3, :3], initial_pose[:3, 3:]
    search = c1
    trace = []
    for i in range(n_iter):
        off = i % 5
        td = depth_img[rows + off, cols + off][:, None]
        cam = rays_dir[rows + off, cols + off, :] * td
        pst7 = pose_6d_to_7d(pst * search)
        world, _, _ = particle_points(rot, trans, pst7, cam)
        mms = mean_masked_sdf(run_network, world, td, trunc)
        rot, trans, search, info = swarm_update(mms, pst7, rot, trans, c2)
        info.update(mean_masked=mms, rot=rot, trans=trans, search=search)
        trace.append(info)
    pose = torch.eye(4, dtype=rot.dtype)
    pose[:3, :3] = rot
    pose[:3, 3:] = trans
    return pose, trace
