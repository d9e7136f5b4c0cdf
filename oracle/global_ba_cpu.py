"""CPU restatement of the reference's neural cross-sub-map global BA (InactiveMap.global_BA_overlapping,
InactiveMap.py:375-474; get_SDF_dif / get_SDF_dif2 / infer_pts, InactiveMap.py:128-192;
compute_avg_SDF_difference, helper_functions/geometry_helper.py:225-229) as ONE process over a list of sub-map models.

TEST INFRASTRUCTURE ONLY (tests/): the checker of mipsfusion_amd/global_ba.py, never imported by the product.
Pinning: the reference's function is reachable only through the two-process orchestrator and its only call site is
commented out (InactiveMap.py:86), so it cannot be executed as shipped; the arithmetic is restated line by line from
the cited ranges and its building blocks (run_network, quaternion helpers) are the pinned oracle pieces
(path_cpu.CpuScene, p3d_cpu).  Status: restated, not executed against the reference -> "parity unpinned" for this
function as a whole.
"""
import torch

from . import p3d_cpu


def infer_sdf(local_poses, model, rays_d_cam, target_d, trunc):
    """infer_pts (InactiveMap.py:128-139), SDF part."""
    rays_d = torch.sum(rays_d_cam[..., None, None, :] * local_poses[..., None, :3, :3], -1)
    rays_o = local_poses[..., None, :3, -1].repeat(1, rays_d.shape[1], 1).reshape(-1, 3)
    rays_d = rays_d.reshape(-1, 3)
    pts_local = (rays_o[..., None, :] + rays_d[..., None, :] * target_d[..., :, None]).reshape(-1, 3)
    return model.run_network(pts_local)[..., 3:4] * trunc


def sdf_dif(models, rays, kf_pose, i, j, pose_i, pose_j, trunc, mask=None):
    """get_SDF_dif / get_SDF_dif2 (InactiveMap.py:149-192) without the zero-weighted colour term."""
    d_cam, target_d = rays[..., :3], rays[..., 6:7]
    if mask is None:
        mask = torch.where(target_d > 0., torch.ones_like(target_d), torch.zeros_like(target_d))
    s1 = infer_sdf(pose_i.inverse() @ kf_pose, models[i], d_cam, target_d, trunc)
    s2 = infer_sdf(pose_j.inverse() @ kf_pose, models[j], d_cam, target_d, trunc)
    loss = torch.sum(torch.square(s1 * mask - s2 * mask))
    return loss / (torch.count_nonzero(mask) + 0.001)


def optimise(models, first_kf_poses, batches, trunc, lr_rot=1e-3, lr_trans=1e-3, pose_accum_step=1):
    """batches: list over iterations of lists of (i, j, rays [N,7], kf_pose_world [N,4,4] or [1,4,4], weight, mask|None).
    -> (anchor poses [n,4,4] after the loop, loss trace).  Loop of InactiveMap.py:395-459."""
    fixed = first_kf_poses[:1].clone()
    cur_trans = torch.nn.Parameter(first_kf_poses[1:, :3, 3].clone())
    cur_rot = torch.nn.Parameter(p3d_cpu.matrix_to_quaternion(first_kf_poses[1:, :3, :3]))
    opt = torch.optim.Adam([{"params": cur_rot, "lr": lr_rot}, {"params": cur_trans, "lr": lr_trans}])
    opt.zero_grad()

    def all_poses():
        n = cur_rot.shape[0]
        T = torch.eye(4)[None].repeat(n, 1, 1)
        T[:, :3, :3] = p3d_cpu.quaternion_to_matrix(cur_rot)
        T[:, :3, 3] = cur_trans
        return torch.cat([fixed, T], 0)

    trace = []
    poses_all = all_poses()
    for it, terms in enumerate(batches):
        loss = 0.
        for (i, j, rays, kf_pose, weight, mask) in terms:
            loss = loss + weight * sdf_dif(models, rays, kf_pose, i, j, poses_all[i], poses_all[j], trunc, mask)
        loss.backward(retain_graph=True)
        trace.append(float(loss.detach()))
        if (it + 1) % pose_accum_step == 0:
            opt.step()
            poses_all = all_poses()
            opt.zero_grad()
    return all_poses().detach(), trace
