"""Measured tracking + mapping time per frame on a synthetic trajectory (BASELINE config 3 in miniature).

A camera moves on an arc inside the box room of mipsfusion_amd/synth.py.  Every frame is tracked the way
MIPSFusion.tracking does it (mipsfusion.py:456-575): constant-velocity prediction -> RandomOptimizer.optimize
(iter_RO rounds) -> `tracking.iter` pose-only Adam iterations on `tracking.sample` rays with the map frozen; every
`keyframe_every`-th frame stores its down-sampled rays in the device-resident keyframe database; every
`map_every`-th frame runs `mapping.iters` local-BA iterations (keyframe rays + current-frame rays, pose + map Adam).
Only product modules are used (JointEncoding, RandomOptimizer, DeviceRayDB, FusedAdam, pose_rays).  Prints one
JSON line: mean / median ms per frame (wall clock, eager launches, device synchronised per frame), the split into
RO / GO / BA, and the trajectory error against the synthetic ground truth.

``--graph`` runs the same cadence with the iterations replayed as hipGraphs (mipsfusion_amd/sequence.py): all tracking
iterations of a frame are one replay, all mapping iterations of a BA round another; rays are gathered inside the graph
from one device table (keyframe database + current frame) through static index / jitter buffers.  ``--sampler
reference`` (default) fills them from the reference's own host generators, run one mapping period ahead by producer threads
(bit-identical index stream); ``--sampler device`` draws indices and jitter on the GPU.

usage: python tools/run_sequence.py [--frames 60] [--rays 4096] [--graph [--sampler reference|device]]
"""
import argparse
import json
import os
import random
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def _usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


import importlib.util  # noqa: E402  (by path: importing the package would import torch before OMP_NUM_THREADS is set)

_spec = importlib.util.spec_from_file_location("mipsf_hostcpu", os.path.join(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__))), "mipsfusion_amd", "hostcpu.py"))
_hostcpu = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_hostcpu)
if not os.environ.get("MIPSF_NO_CONFINE"):
    _hostcpu.confine_to_numa_node(32)
os.environ.setdefault("OMP_NUM_THREADS", str(_usable_cores()))      # see bench.py: the hosts report 256 cores, grant 16
import torch  # noqa: E402

from mipsfusion_amd import ops, synth  # noqa: E402
from mipsfusion_amd.RandomOptimizer import RandomOptimizer  # noqa: E402
from mipsfusion_amd.helper_functions import sampling_helper as sh  # noqa: E402
from mipsfusion_amd.helper_functions.geometry_helper import matrix_to_quaternion, qt_to_transform_matrix  # noqa: E402
from mipsfusion_amd.helper_functions.utils import get_loss_from_ret  # noqa: E402
from mipsfusion_amd.keyframe_rays import DeviceRayDB  # noqa: E402
from mipsfusion_amd.model import JointEncoding  # noqa: E402
from mipsfusion_amd.optim import FusedAdam  # noqa: E402


def trajectory(cfg, n):
    poses = []
    for k in range(n):
        a = k / max(1, n - 1)
        c2w = synth.default_pose(cfg, yaw=0.25 + 0.5 * a, pitch=-0.1 + 0.05 * np.sin(6.28 * a))
        c2w[:3, 3] += torch.tensor([0.5 * a, 0.8 * a, 0.1 * np.sin(3.14 * a)], dtype=c2w.dtype)
        poses.append(c2w)
    return poses


def frame_rays(frame):
    """[H*W, 7] rows [direction | rgb | depth] of one frame (mipsfusion.py:296-297)."""
    return torch.cat([frame["direction"], frame["rgb"], frame["depth"][..., None]], -1).reshape(-1, 7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--rays", type=int, default=4096, help="rays per local-BA iteration")
    ap.add_argument("--first-iters", type=int, default=200, help="first-frame mapping iterations (reference: 500)")
    ap.add_argument("--profile", action="store_true", help="cProfile the per-frame loop (host side)")
    ap.add_argument("--graph", action="store_true", help="replay captured iterations instead of launching eagerly")
    ap.add_argument("--sampler", choices=("reference", "device"), default="reference",
                    help="--graph only: the reference's host generators run ahead in producer threads, or device draws")
    a = ap.parse_args()
    if a.graph:
        return main_graphed(a)
    dev = torch.device("cuda:0")
    random.seed(0), np.random.seed(0), torch.manual_seed(0)
    cfg = synth.config_headline()
    tr, mp = cfg["tracking"], cfg["mapping"]
    tr["RO"].setdefault("initial_scaling_factor", 0.02)
    tr["RO"].setdefault("rescaling_factor", 0.5)
    tr.setdefault("ignore_edge_W", 20), tr.setdefault("ignore_edge_H", 20)
    kf_every, map_every = 15, mp["map_every"]
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    bb = torch.from_numpy(np.array(mp["bound"]))
    nf = torch.from_numpy(np.array(mp["localMLP_max_len"]))
    model = JointEncoding(cfg, bb, nf).to(dev).train()
    model.accumulate_param_grads_in_place = True
    gt = trajectory(cfg, a.frames)
    frames = [synth.make_frame(cfg, gt[k], seed=k, frame_id=k) for k in range(a.frames)]
    ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frames[0]["direction"])
    ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
    kf_rows, kf_cols = sh.sample_pixels_uniformly(H, W, 100, 300 if W >= 300 else W // 2)    # 30 000 rays per keyframe
    n_save = kf_rows.shape[0]
    db = DeviceRayDB(a.frames // kf_every + 2, n_save, dev)
    kf_ids, kf_pose_q, kf_pose_t = [], [], []
    map_opt = FusedAdam([{"params": model.decoder.parameters(), "weight_decay": 1e-6, "lr": mp["lr_decoder"]},
                         {"params": model.embed_fn.parameters(), "eps": 1e-15, "lr": mp["lr_embed"]}], betas=(0.9, 0.99))

    def add_keyframe(k, pose):
        db.store(len(kf_ids), frame_rays(frames[k]).reshape(H, W, 7)[kf_rows, kf_cols].to(dev))
        kf_ids.append(k)
        kf_pose_q.append(matrix_to_quaternion(pose[None, :3, :3].to(dev))[0])
        kf_pose_t.append(pose[:3, 3].to(dev).clone())

    def local_ba(cur_rays_dev, cur_pose, iters):
        """mipsfusion.py:293-342: keyframe rays + current-frame rays, pose (all but the first keyframe) + map Adam."""
        K = len(kf_ids)
        rot = torch.nn.Parameter(torch.stack(kf_pose_q[1:] + [matrix_to_quaternion(cur_pose[None, :3, :3].to(dev))[0]]))
        trans = torch.nn.Parameter(torch.stack(kf_pose_t[1:] + [cur_pose[:3, 3].to(dev)]))
        fixed = qt_to_transform_matrix(kf_pose_q[0][None], kf_pose_t[0][None]).detach()
        popt = FusedAdam([{"params": rot, "lr": mp["lr_rot"]}, {"params": trans, "lr": mp["lr_trans"]}])
        n_cur = max(a.rays // max(K, 1), mp["pixels_cur"])
        n_kf = a.rays - n_cur
        related = torch.arange(K)
        for i in range(iters):
            rays, _, kf_indices = db.sample_rays_in_submap(torch.tensor(0), related, n_kf)
            idx_cur = torch.tensor(random.sample(range(H * W), n_cur), device=dev)
            batch = torch.cat([rays, cur_rays_dev[idx_cur]], 0)
            owner = torch.cat([kf_indices, torch.full((n_cur,), -1, dtype=torch.int64)]).to(dev)
            rays_o, rays_d = ops.pose_rays(rot, trans, fixed, owner, batch[:, :3].contiguous())
            ret = model.forward(rays_o, rays_d, batch[:, 3:6].contiguous(), batch[:, 6:7].contiguous(),
                                noise=torch.rand(a.rays, S, device=dev))      # jitter drawn on the device
            get_loss_from_ret(ret, cfg["training"]).backward()
            map_opt.step(zero_grad=True)
            if (i + 1) % mp["pose_accum_step"] == 0:
                popt.step(zero_grad=True)
        with torch.no_grad():
            for j in range(1, K):
                kf_pose_q[j], kf_pose_t[j] = rot[j - 1].detach().clone(), trans[j - 1].detach().clone()
            return qt_to_transform_matrix(rot[-1:].detach(), trans[-1:].detach())[0]

    def track(k, init):
        """mipsfusion.py:456-575: RandomOptimizer rounds, then pose-only Adam with the map frozen."""
        f = frames[k]
        t0 = time.perf_counter()
        model.eval()
        pose = ro.optimize(model, f["depth"], init, None, n_iter=tr["iter_RO"])
        model.train()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ro_ms = (t1 - t0) * 1e3
        rot = torch.nn.Parameter(matrix_to_quaternion(pose[None, :3, :3]))
        trans = torch.nn.Parameter(pose[None, :3, 3].clone())
        popt = FusedAdam([{"params": rot, "lr": tr["lr_rot"]}, {"params": trans, "lr": tr["lr_trans"]}])
        rays = frame_rays(f)
        idx = sh.sample_valid_pixels_random(f["depth"][tr["ignore_edge_H"]:-tr["ignore_edge_H"], tr["ignore_edge_W"]:-tr["ignore_edge_W"]], tr["sample"])
        hh, ww = H - 2 * tr["ignore_edge_H"], W - 2 * tr["ignore_edge_W"]
        rr, cc = idx // ww + tr["ignore_edge_H"], idx % ww + tr["ignore_edge_W"]
        b = rays.reshape(H, W, 7)[rr, cc].to(dev)
        own = torch.zeros(b.shape[0], dtype=torch.int64, device=dev)
        for prm in model.parameters():
            prm.requires_grad_(False)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(tr["iter"]):
            rays_o, rays_d = ops.pose_rays(rot, trans, None, own, b[:, :3].contiguous())
            ret = model.forward(rays_o, rays_d, b[:, 3:6].contiguous(), b[:, 6:7].contiguous(), EMD_w=0.,
                                noise=torch.rand(b.shape[0], S, device=dev))
            get_loss_from_ret(ret, cfg["training"]).backward()
            popt.step(zero_grad=True)
        for prm in model.parameters():
            prm.requires_grad_(True)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        out = qt_to_transform_matrix(rot.detach(), trans.detach())[0]
        return out, ro_ms, (t3 - t2) * 1e3, (t2 - t1) * 1e3

    # first frame: ground-truth pose, long mapping (mipsfusion.py:155-194)
    est = [gt[0].to(dev).float()]
    add_keyframe(0, gt[0])
    local_ba(frame_rays(frames[0]).to(dev), gt[0], a.first_iters)
    torch.cuda.synchronize()
    t_frame, t_ro, t_go, t_ba, t_prep = [], [], [], [], []
    prof = None
    if a.profile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for k in range(1, a.frames):
        t0 = time.perf_counter()
        prev = est[-1]
        init = prev if len(est) < 2 else prev @ torch.linalg.inv(est[-2]) @ prev       # constant velocity
        pose, ro_ms, go_ms, prep_ms = track(k, init)
        t_prep.append(prep_ms)
        ba_ms = 0.0
        if k % kf_every == 0:
            add_keyframe(k, pose.cpu())
        if k % map_every == 0:
            tb = time.perf_counter()
            pose = local_ba(frame_rays(frames[k]).to(dev), pose, mp["iters"])
            torch.cuda.synchronize()
            ba_ms = (time.perf_counter() - tb) * 1e3
        est.append(pose.detach())
        torch.cuda.synchronize()
        t_frame.append((time.perf_counter() - t0) * 1e3), t_ro.append(ro_ms), t_go.append(go_ms), t_ba.append(ba_ms)
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
    err = [float((est[k][:3, 3].cpu() - gt[k][:3, 3].float()).norm()) for k in range(a.frames)]
    print(json.dumps({
        "frames": a.frames, "rays_per_ba_iter": a.rays, "ms_per_frame_mean": round(float(np.mean(t_frame)), 3),
        "ms_per_frame_median": round(float(np.median(t_frame)), 3),
        "ro_ms_mean": round(float(np.mean(t_ro)), 3), "go_ms_mean": round(float(np.mean(t_go)), 3),
        "ba_ms_mean_over_all_frames": round(float(np.mean(t_ba)), 3),
        "tracking_host_prep_ms_mean": round(float(np.mean(t_prep)), 3),
        "hot_path_ms_per_frame_median": round(float(np.median(np.array(t_ro) + np.array(t_go) + np.array(t_ba))), 3),
        "hot_path_ms_per_frame_mean": round(float(np.mean(np.array(t_ro) + np.array(t_go) + np.array(t_ba))), 3),
        "cadence": {"iter_RO": tr["iter_RO"], "tracking_iter": tr["iter"], "mapping_iters": mp["iters"],
                    "map_every": map_every, "keyframe_every": kf_every},
        "launch": "eager; includes the host-side pixel / keyframe-ray index sampling (python RNG, as in the reference) "
                  "and the per-iteration index upload; sample jitter is drawn on the device",
        "ate_rmse_m": round(float(np.sqrt(np.mean(np.square(err)))), 4), "ate_max_m": round(max(err), 4),
        "trajectory_length_m": round(float(sum((gt[k][:3, 3] - gt[k - 1][:3, 3]).norm() for k in range(1, a.frames))), 3)}))


def main_graphed(a):
    """hipGraph replays + host sampling off the critical path: mipsfusion_amd/sequence.py."""
    from mipsfusion_amd import sequence
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    random.seed(0), np.random.seed(0), torch.manual_seed(0)
    cfg = synth.config_headline()
    if a.rays != 4096:
        cfg["mapping"]["sample"] = a.rays - cfg["mapping"]["pixels_cur"]
    gt = trajectory(cfg, a.frames)
    frames = [synth.make_frame(cfg, gt[k], seed=k, frame_id=k) for k in range(a.frames)]
    seq = sequence.GraphedSequence(cfg, dev, frames, kf_every=15, sampler=a.sampler, first_iters=a.first_iters)
    res = seq.run(gt)
    launch = ("hipGraph replay per tracking frame / per BA round; RandomOptimizer rounds of a frame in one replay; includes the 8 MB pinned frame upload; "
              + ("pixel / keyframe-ray indices and jitter from the reference's host generators (python random, torch CPU), "
                 "drawn map_every frames ahead by producer threads" if a.sampler == "reference" else
                 "indices + jitter drawn on the device (valid depth only, without replacement)"))
    out = sequence.summarise(res, gt, cfg, launch)
    print("per-frame ms (RO, GO, BA, wait):", " ".join(f"{r:.1f}/{g:.1f}/{b:.1f}/{w:.1f}" for r, g, b, w in
                                                        zip(res["ro_ms"], res["go_ms"], res["ba_ms"], res["producer_wait_ms"])),
          file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
