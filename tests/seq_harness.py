"""BASELINE config 3 in miniature: the online loop of MIPSFusion.run (mipsfusion.py:661-735) over a short synthetic
sequence with TWO sub-maps -- first-frame mapping, per-frame tracking (RandomOptimizer rounds + pose-only Adam),
local BA every ``map_every`` frames, a switch to a NEW sub-map (``active_submap_switch_new`` + ``initialize_new_localMLP``,
mipsfusion.py:198-222, 637-652) and a switch BACK to the first one (``active_submap_switch`` + ``local_BA_switch``,
mipsfusion.py:379-444, 608-634).

TEST INFRASTRUCTURE.  The loop is written once against a small ``Backend`` so that the very same control flow --
and therefore the very same sequence of python-``random`` / torch-CPU / numpy RNG calls -- runs

  * in the build container over the REFERENCE's own classes (``model.scene_rep.JointEncoding``, ``KeyframeSet``,
    ``RandomOptimizer``, ``sampling_helper``; tests/golden/make_golden.py::gen_sequence -> tests/golden/sequence.npz), and
  * on the GPU box over the product's modules (tests/test_gpu_sequence.py),

and the two runs can be compared: index stream bit for bit, loss trace / poses / weights within tolerance.

What is simplified with respect to the reference (all of it control plane, SURVEY section 2 "OUT OF SCOPE"): the
sub-map decision logic of Manager.py is replaced by fixed switch frames, the two-process hand-off by in-process
``deepcopy`` / ``load_state_dict`` calls in the same order (InactiveMap.py:66-70, 81-88; mipsfusion.py:616, 632, 642),
PoseCorrector's ICP rectification by the anchor-pose composition of ``current_pose_switch_submap``
(mipsfusion.py:587-603), the overlap mutex bookkeeping is dropped (one process).
"""
import copy
import random
import types

import numpy as np
import torch


def sequence_config():
    """Config-1 sized frames (32x32, hash 2^10, S = 11 + 5) with the reference's online cadence scaled down."""
    from mipsfusion_amd import synth
    cfg = synth.config_plumbing()
    cfg["sampling"] = {"n_rays_h": 4, "n_rays_w": 6, "kf_n_rays_h": 12, "kf_n_rays_w": 16}
    cfg["mapping"].update(sample=96, pixels_cur=40, iters=3, first_iters=6, keyframe_every=2, map_every=2,
                          map_accum_step=1, pose_accum_step=2, map_wait_step=0, optim_cur=False, localMLP_num=2,
                          overlapping={"n_rays_h": 2, "n_rays_w": 2})
    cfg["tracking"].update(iter_RO=2, iter=3, sample=64, wait_iters=100, const_speed=True, best=True,
                           ignore_edge_W=2, ignore_edge_H=2,
                           switch={"lr_rot": 0.001, "lr_trans": 0.001, "map_num": 3})
    cfg["tracking"]["RO"] = {"particle_size": 96, "n_rows": 4, "n_cols": 6, "initial_scaling_factor": 0.02,
                             "rescaling_factor": 0.5}
    return cfg


def trajectory(cfg, n):
    from mipsfusion_amd import synth
    poses = []
    for k in range(n):
        a = k / max(1, n - 1)
        c2w = synth.default_pose(cfg, yaw=0.25 + 0.25 * a, pitch=-0.1 + 0.03 * a)
        c2w[:3, 3] += torch.tensor([0.10 * a, 0.06 * a, 0.02 * a])
        poses.append(c2w)
    return poses


def total_loss(ret, tr):
    """MIPSFusion.get_loss_from_ret (mipsfusion.py:142-152)."""
    return (tr["rgb_weight"] * ret["rgb_loss"] + tr["depth_weight"] * ret["depth_loss"]
            + tr["sdf_weight"] * ret["sdf_loss"] + tr["fs_weight"] * ret["fs_loss"])


class Recorder:
    def __init__(self):
        self.idx, self.loss, self.tags = [], [], []

    def index(self, tag, t):
        self.idx.append(torch.as_tensor(t).detach().cpu().to(torch.int64).reshape(-1).clone())
        self.tags.append(tag)

    def add_loss(self, v):
        self.loss.append(float(v.detach()) if torch.is_tensor(v) else float(v))


def run_sequence(B, n_frames=9, switch_new=4, switch_back=8, seed=0):
    """B: backend namespace (see make_golden.reference_backend / test_gpu_sequence.product_backend).
    -> dict(idx=[...], tags=[...], losses=[...], est=[F,4,4] local poses, models={submap: state_dict})."""
    from mipsfusion_amd import synth
    random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
    cfg = sequence_config()
    tr, mp, tk = cfg["training"], cfg["mapping"], cfg["tracking"]
    dev = B.device
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    gt = trajectory(cfg, n_frames)
    frames = [synth.make_frame(cfg, gt[k], seed=100 + k, frame_id=k) for k in range(n_frames)]
    bb = torch.from_numpy(np.array(mp["bound"]))
    nf = torch.from_numpy(np.array(mp["localMLP_max_len"]))
    # the sub-map frame is its first keyframe's camera: shift the bound so the camera-centred scene stays inside
    model = B.make_model(cfg, bb - bb.mean(1, keepdim=True), nf)
    shared_model = B.deepcopy(model)                     # mipsfusion.py:112-113 (the hand-off buffer)
    active_model_copy = B.deepcopy(model)                # InactiveMap's copy, refreshed after every BA (mipsfusion.py:683)
    model_list = {}                                      # InactiveMap.model_list (InactiveMap.py:28, 66-70)
    rec = Recorder()

    def new_map_optimizer():                             # mipsfusion.py:580-584
        return B.Adam([{"params": model.decoder.parameters(), "weight_decay": 1e-6, "lr": mp["lr_decoder"]},
                       {"params": model.embed_fn.parameters(), "eps": 1e-15, "lr": mp["lr_embed"]}], betas=(0.9, 0.99))

    def pose_params(poses, lr_rot, lr_trans):            # mipsfusion.py:235-250
        cur_trans = torch.nn.Parameter(poses[:, :3, 3].clone())
        cur_rot = torch.nn.Parameter(B.matrix_to_quaternion(poses[:, :3, :3]))
        return cur_rot, cur_trans, B.Adam([{"params": cur_rot, "lr": lr_rot}, {"params": cur_trans, "lr": lr_trans}])

    map_opt = new_map_optimizer()
    kfs = B.make_kfset(cfg, H, W, n_frames // mp["keyframe_every"] + 2)
    dataset = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frames[0]["direction"])
    ro = B.make_ro(cfg, types.SimpleNamespace(dataset=dataset, device=dev))

    est = torch.eye(4)[None].repeat(n_frames, 1, 1).to(dev)      # est_c2w_data: LOCAL pose of every frame
    submap_kfs = {0: []}                                         # submap -> [(kf_id, frame_id)], first entry = first kf
    anchor = {0: torch.eye(4, device=dev)}                       # pose of each sub-map's frame in sub-map 0's frame
    active, last_switch, optim_cur = 0, 0, mp["optim_cur"]
    n_kf = 0

    def init_iterations(frame, n_iters):
        """first_frame_mapping / initialize_new_localMLP body (mipsfusion.py:172-190, 206-221), quirk included:
        rows are `indice % H` and columns `indice // H`."""
        model.train()
        c2w_local = torch.eye(4, device=dev)
        for _ in range(n_iters):
            map_opt.zero_grad()
            indice = B.sh.sample_pixels_random(H, W, mp["sample"])        # = MIPSFusion.select_samples
            rec.index("init", indice)
            ih, iw = torch.remainder(indice, H), torch.div(indice, H, rounding_mode="floor")
            d_cam = frame["direction"][ih, iw, :].to(dev)
            t_s, t_d = frame["rgb"][ih, iw, :].to(dev), frame["depth"][ih, iw].to(dev).unsqueeze(-1)
            rays_o = c2w_local[None, :3, -1].repeat(mp["sample"], 1)
            rays_d = torch.sum(d_cam[..., None, :] * c2w_local[:3, :3], -1)
            ret = model.forward(rays_o, rays_d, t_s, t_d)
            loss = total_loss(ret, tr)
            loss.backward()
            map_opt.step()
            rec.add_loss(loss)

    def tracking(i, switch_tracking=False):
        """tracking_render (mipsfusion.py:470-577)."""
        f = frames[i]
        if switch_tracking:
            cur = est[i].clone()
        elif tk["const_speed"] and (i - last_switch) >= 2:
            cur = (est[i - 1] @ torch.linalg.inv(est[i - 2])) @ est[i - 1]
        else:
            cur = est[i - 1].clone()
        if tk["iter_RO"] > 0:
            last_pose = est[i - 1].clone()
            cur = B.ro_optimize(ro, model, f["depth"], cur.clone(), last_pose, tk["iter_RO"]).to(dev)
        cur_rot, cur_trans, popt = pose_params(cur[None], tk["lr_rot"], tk["lr_trans"])
        rows = cols = None
        best_loss, best_pose, thresh = None, None, 0
        for _ in range(tk["iter"]):
            popt.zero_grad()
            c2w_est = B.qt_to_transform_matrix(cur_rot, cur_trans)
            if rows is None:
                rows, cols = B.sh.sample_pixels_mix(H, W, cfg["sampling"]["n_rays_h"], cfg["sampling"]["n_rays_w"],
                                                    f["depth"], tk["sample"])
                rec.index("track", B.sh.pixel_rc_to_indices(rows, cols, H, W))
                d_cam = f["direction"][rows, cols, :].to(dev)
                t_s, t_d = f["rgb"][rows, cols, :].to(dev), f["depth"][rows, cols].to(dev).unsqueeze(-1)
            rays_o = c2w_est[..., :3, -1].repeat(tk["sample"], 1)
            rays_d = torch.sum(d_cam[..., None, :] * c2w_est[:, :3, :3], -1)
            ret = model.forward(rays_o, rays_d, t_s, t_d, EMD_w=0.)
            loss = total_loss(ret, tr)
            rec.add_loss(loss)
            lv = float(loss.detach())
            if best_loss is None:                               # mipsfusion.py:540-553, literally
                best_loss, best_pose = lv, c2w_est.detach()
            if lv < best_loss:
                best_loss, best_pose, thresh = lv, c2w_est.detach(), 0
            else:
                thresh += 1
            if thresh > tk["wait_iters"]:
                break
            loss.backward()
            popt.step()
        est[i] = best_pose.detach().clone()[0] if tk["best"] else c2w_est.detach().clone()[0]

    def local_ba(i):
        """local_BA (mipsfusion.py:259-371) for the active sub-map; sub-map bookkeeping reduced to `submap_kfs`."""
        f = frames[i]
        rel = submap_kfs[active]
        kf_ids_all = torch.tensor([k for k, _ in rel])
        frame_ids_all = [fid for _, fid in rel]
        first_kf_Id = kf_ids_all[0]
        poses = est[frame_ids_all].clone()
        poses[0] = torch.eye(4, device=dev)
        popt = None
        current_pose = est[i][None, ...]
        if len(kf_ids_all) < 2:
            poses_fixed = poses
            poses_all = torch.cat([poses_fixed, current_pose], 0)
        else:
            poses_fixed = poses[:1]
            if optim_cur:
                cur_rot, cur_trans, popt = pose_params(torch.cat([poses[1:], current_pose]), mp["lr_rot"], mp["lr_trans"])
                poses_all = torch.cat([poses_fixed, B.qt_to_transform_matrix(cur_rot, cur_trans)], 0)
            else:
                cur_rot, cur_trans, popt = pose_params(poses[1:], mp["lr_rot"], mp["lr_trans"])
                poses_all = torch.cat([poses_fixed, B.qt_to_transform_matrix(cur_rot, cur_trans), current_pose], 0)
        map_opt.zero_grad()
        if popt is not None:
            popt.zero_grad()
        cur_raw = torch.cat([f["direction"], f["rgb"], f["depth"][..., None]], -1)            # [H,W,7]
        for it in range(mp["iters"]):
            rays, kf_ids, kf_indices = kfs.sample_rays_in_submap(first_kf_Id, kf_ids_all, mp["sample"])
            rec.index("ba_kf", kf_indices)
            n_cur = max(mp["sample"] // kf_ids_all.shape[0], mp["pixels_cur"])
            rows, cols = B.sh.sample_pixels_mix(H, W, tk["RO"]["n_rows"], tk["RO"]["n_cols"], f["depth"], n_cur)
            rec.index("ba_cur", B.sh.pixel_rc_to_indices(rows, cols, H, W))
            rays = torch.cat([rays.to(dev), cur_raw[rows, cols].to(dev)], 0)
            rec.index("ba_rays_checksum", (rays[:, 6] * 1e4).round())       # which rows came out of the ray database
            indices_all = torch.cat([kf_indices.to(dev), -torch.ones((n_cur,), device=dev)]).to(torch.int64)
            d_cam, t_s, t_d = rays[..., :3], rays[..., 3:6], rays[..., 6:7]
            rays_d = torch.sum(d_cam[..., None, None, :] * poses_all[indices_all, None, :3, :3], -1)
            rays_o = poses_all[indices_all, :3, -1].repeat(1, rays_d.shape[1], 1).reshape(-1, 3)
            rays_d = rays_d.reshape(-1, 3)
            ret = model.forward(rays_o, rays_d, t_s.contiguous(), t_d.contiguous())
            loss = total_loss(ret, tr)
            rec.add_loss(loss)
            loss.backward(retain_graph=True)
            if (it + 1) % mp["map_accum_step"] == 0:
                if (it + 1) > mp["map_wait_step"]:
                    map_opt.step()
                map_opt.zero_grad()
            if popt is not None and (it + 1) % mp["pose_accum_step"] == 0:
                popt.step()
                pose_optim = B.qt_to_transform_matrix(cur_rot, cur_trans)
                poses_all = torch.cat([poses_fixed, pose_optim] + ([] if optim_cur else [current_pose]), 0)
                popt.zero_grad()
        if popt is not None and len(kf_ids_all) > 1:
            with torch.no_grad():
                for j in range(len(kf_ids_all) - 1):
                    est[frame_ids_all[1:][j]] = B.qt_to_transform_matrix(cur_rot[j:j + 1], cur_trans[j:j + 1]).detach().clone()[0]
                if optim_cur:
                    est[i] = B.qt_to_transform_matrix(cur_rot[-1:], cur_trans[-1:]).detach().clone()[0]

    def local_ba_switch(i, given):
        """local_BA_switch (mipsfusion.py:379-444): pose-only refinement of the overlapping keyframe after a switch
        back; the map gradients accumulate in .grad and are never stepped (cleared by the next local BA)."""
        f = frames[i]
        kf_ids_all = torch.tensor([k for k, _ in given])
        poses = est[[fid for _, fid in given]].clone()
        poses[0] = torch.eye(4, device=dev)
        poses_fixed = poses
        cur_rot, cur_trans, popt = pose_params(est[i].detach()[None], tk["switch"]["lr_rot"], tk["switch"]["lr_trans"])
        poses_all = torch.cat([poses_fixed, B.qt_to_transform_matrix(cur_rot, cur_trans)], 0)
        popt.zero_grad()
        ovlp = torch.cat([f["direction"], f["rgb"], f["depth"][..., None]], -1).reshape(-1, 7)
        for it in range(tk["switch"]["map_num"]):
            n_ov = max(mp["sample"] // kf_ids_all.shape[0], mp["sample"] // 5)
            rays, kf_ids, kf_indices = kfs.sample_rays_in_given_kf(kf_ids_all, mp["sample"])
            rec.index("sw_kf", kf_indices)
            idx_cur = random.sample(range(0, H * W), n_ov)
            rec.index("sw_cur", torch.tensor(idx_cur))
            rays = torch.cat([rays.to(dev), ovlp[idx_cur, :].to(dev)], 0)
            indices_all = torch.cat([kf_indices.to(dev), -torch.ones((n_ov,), device=dev)]).to(torch.int64)
            d_cam, t_s, t_d = rays[..., :3], rays[..., 3:6], rays[..., 6:7]
            rays_d = torch.sum(d_cam[..., None, None, :] * poses_all[indices_all, None, :3, :3], -1)
            rays_o = poses_all[indices_all, None, :3, -1].repeat(1, rays_d.shape[1], 1).reshape(-1, 3)
            rays_d = rays_d.reshape(-1, 3)
            ret = model.forward(rays_o, rays_d, t_s.contiguous(), t_d.contiguous())
            loss = total_loss(ret, tr)
            rec.add_loss(loss)
            loss.backward(retain_graph=True)
            if (it + 1) % mp["pose_accum_step"] == 0:
                popt.step()
                poses_all = torch.cat([poses_fixed, B.qt_to_transform_matrix(cur_rot, cur_trans)], 0)
                popt.zero_grad()
        est[i] = B.qt_to_transform_matrix(cur_rot, cur_trans).detach().clone()[0]

    # ------------------------------------------------------------------ the loop (mipsfusion.py:674-723)
    init_iterations(frames[0], mp["first_iters"])
    kfs.add_keyframe(frames[0])
    submap_kfs[0].append((0, 0))
    n_kf = 1
    for i in range(1, n_frames):
        tracking(i)
        if i % mp["map_every"] == 0:
            local_ba(i)
            active_model_copy.load_state_dict(model.state_dict())                 # mipsfusion.py:683
        if i % mp["keyframe_every"] == 0:
            kf_id = i // mp["keyframe_every"]
            kfs.add_keyframe(frames[i])
            n_kf += 1
            if i == switch_new:                                                    # return_flag == 3
                shared_model.load_state_dict(model.state_dict())                   # mipsfusion.py:642
                model_list[active] = B.deepcopy(shared_model)                      # InactiveMap.py:66-70
                model.recover_initial_param()                                      # mipsfusion.py:648
                anchor[1] = anchor[active] @ est[i]
                prev, active = active, 1
                submap_kfs[1] = [(kf_id, i)]
                last_switch = i
                est[i] = torch.eye(4, device=dev)
                map_opt = new_map_optimizer()                                      # initialize_new_localMLP: create_optimizer
                init_iterations(frames[i], mp["first_iters"])
            elif i == switch_back:                                                 # return_flag == 1
                shared_model.load_state_dict(model.state_dict())                   # mipsfusion.py:616
                model_list[active] = B.deepcopy(shared_model)                      # InactiveMap.py:81
                shared_model.load_state_dict(model_list[0].state_dict())           # InactiveMap.py:84-88
                # current_pose_switch_submap (mipsfusion.py:587-603): local pose in the sub-map switched to
                est[i] = torch.linalg.inv(anchor[0]) @ (anchor[active] @ est[i])
                model.load_state_dict(shared_model.state_dict())                   # mipsfusion.py:632
                active, last_switch, optim_cur = 0, i, True                        # mipsfusion.py:634
                local_ba_switch(i, submap_kfs[0])
                submap_kfs[0].append((kf_id, i))
            else:
                submap_kfs[active].append((kf_id, i))
    model_list[active] = B.deepcopy(model)
    out = {"idx": rec.idx, "tags": rec.tags, "losses": np.array(rec.loss), "est": est.detach().cpu().numpy(),
           "models": {k: {n: v.detach().cpu().clone() for n, v in m.state_dict().items()} for k, m in model_list.items()},
           "active_copy": {n: v.detach().cpu().clone() for n, v in active_model_copy.state_dict().items()}}
    return out
