"""GPU (-m gpu): the BASELINE configurations at their full sizes against the CPU oracle.

  config 2   one complete mapping iteration at 4096 rays x 64 samples, hash 2^19, EMD on: every returned tensor, the
             grid / decoder / ray gradients and the parameters after one FusedAdam step vs oracle/path_cpu.py
  config 5   ScanNet scene0000 workload (scene bound, floor-divided intrinsics, far 7, S = 75): a tracking iteration
             with the reference's iter_RO = 0 pixel sampling, and the RandomOptimizer forward-only slice [2000, 384]

Two error metrics are reported for every floating-point comparison: the error relative to the
tensor's maximum magnitude, and the per-element relative error with an absolute floor (|a-b| / max(|b|, floor))."""
import random
import types

import numpy as np
import pytest
import torch

from mipsfusion_amd import ops, synth
from mipsfusion_amd.helper_functions import sampling_helper as sh
from mipsfusion_amd.helper_functions.geometry_helper import matrix_to_quaternion, qt_to_transform_matrix
from mipsfusion_amd.model import JointEncoding
from mipsfusion_amd.optim import FusedAdam
from oracle import path_cpu, ro_cpu

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    return torch.device("cuda:0")


def _np(a):
    return a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)


def both_errors(a, b, floor):
    """-> (max |a-b| / max |b|,  max |a-b| / max(|b|, floor))."""
    a, b = _np(a), _np(b)
    d = np.abs(a - b)
    return float(d.max() / (np.abs(b).max() + 1e-30)), float((d / np.maximum(np.abs(b), floor)).max())


def check(a, b, what, tol_max=1e-4, tol_elem=1e-4, floor=1e-3):
    e_max, e_elem = both_errors(a, b, floor)
    print(f"  {what:34s} rel-to-max {e_max:.2e}   per-element (floor {floor:g}) {e_elem:.2e}")
    assert e_max <= tol_max, f"{what}: error relative to max magnitude {e_max:.3e} > {tol_max:.1e}"
    assert e_elem <= tol_elem, f"{what}: per-element relative error {e_elem:.3e} > {tol_elem:.1e} (floor {floor:g})"


def check_grad(a, b, what, tol=5e-4, outlier_frac=1e-4):
    """Gradients cross ReLU / first-crossing / band-mask decisions: a few of millions of them legitimately flip
    between two fp32 summation orders.  Bound the relative L2 error and the fraction of entries off by > tol*max."""
    a, b = _np(a).ravel(), _np(b).ravel()
    scale = np.abs(b).max() + 1e-30
    bad = np.abs(a - b) > tol * scale
    l2 = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
    print(f"  {what:34s} relative L2 {l2:.2e}   entries off by > {tol:g}*max: {int(bad.sum())} of {bad.size}")
    assert bad.mean() <= outlier_frac and l2 <= 4 * tol, f"{what}: {bad.sum()} outliers, relative L2 {l2:.3e}"


def build(cfg, dev, seed=0, grid_scale=0.2):
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    torch.manual_seed(seed)
    m = JointEncoding(cfg, bb, nf).to(dev)
    with torch.no_grad():
        m.embed_fn.params.copy_((torch.randn(m.embed_fn.params.shape) * grid_scale).to(dev))
        m.decoder.sdf_linear[2].weight.mul_(3.0)           # sign changes along the rays
    cpu = path_cpu.CpuScene(cfg, bb, nf)
    cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    return m, cpu


def map_groups(model, cfg):
    return [{"params": model.decoder.parameters(), "weight_decay": 1e-6, "lr": cfg["mapping"]["lr_decoder"]},
            {"params": model.embed_fn.parameters(), "eps": 1e-15, "lr": cfg["mapping"]["lr_embed"]}]


# ------------------------------------------------------------------------------------------------ config 2
def test_config2_full_iteration_vs_oracle(dev):
    """BASELINE config 2 in full: 4096 rays x 64 samples (43 + 21), hash 2^19, apartment_2 bound, EMD 0.01, rays built
    from optimisable keyframe poses; forward tensors, all gradients and the parameters after one Adam step."""
    cfg = synth.config_headline()
    assert cfg["grid"]["hash_size"] == 19
    m, cpu = build(cfg, dev)
    m.train()
    N, S = 4096, 64
    # 4 keyframes + the current frame, mapping-style batch (mipsfusion.py:293-322)
    random.seed(0), torch.manual_seed(1)
    poses, recs, owner = [], [], []
    for k in range(4):
        c2w = synth.default_pose(cfg, yaw=0.3 + 0.12 * k, pitch=-0.1 + 0.02 * k)
        c2w[:3, 3] += torch.tensor([0.05 * k, 0.08 * k, 0.0])
        f = synth.make_frame(cfg, c2w, seed=k, frame_id=k)
        H, W = f["depth"].shape
        idx = torch.tensor(random.sample(range(H * W), N // 4))
        r, c = torch.div(idx, W, rounding_mode="floor"), torch.remainder(idx, W)
        recs.append(torch.cat([f["direction"][r, c], f["rgb"][r, c], f["depth"][r, c][:, None]], -1))
        owner.append(torch.full((N // 4,), k, dtype=torch.int64))
        poses.append(c2w)
    rays, owner, poses = torch.cat(recs), torch.cat(owner), torch.stack(poses)
    noise = torch.rand(N, S)

    def pose_leaves(device):
        rot = torch.nn.Parameter(matrix_to_quaternion(poses[1:, :3, :3]).to(device))
        trans = torch.nn.Parameter(poses[1:, :3, 3].clone().to(device))
        return rot, trans

    # ---- product
    rot, trans = pose_leaves(dev)
    rays_o, rays_d = ops.pose_rays(rot, trans, poses[:1].to(dev), owner.to(dev), rays[:, :3].contiguous().to(dev))
    ret = m.forward(rays_o, rays_d, rays[:, 3:6].contiguous().to(dev), rays[:, 6:7].contiguous().to(dev), EMD_w=0.01,
                    noise=noise.to(dev))
    loss = path_cpu.total_loss(ret, cfg["training"])
    loss.backward()
    # ---- oracle (torch CPU, reference ray composition mipsfusion.py:320-322)
    rot_c, trans_c = pose_leaves("cpu")
    poses_all = torch.cat([poses[:1], qt_to_transform_matrix(rot_c, trans_c)], 0)
    rd_c = torch.sum(rays[:, None, None, :3] * poses_all[owner, None, :3, :3], -1).reshape(-1, 3)
    ro_c = poses_all[owner, :3, -1]
    ref = cpu.train_forward(ro_c, rd_c, rays[:, 3:6], rays[:, 6:7], noise, 0.01)
    ref_loss = path_cpu.total_loss(ref, cfg["training"])
    ref_loss.backward()

    print("\nconfig 2, full iteration, 4096 x 64, hash 2^19, EMD 0.01:")
    check(rays_o, ro_c, "rays_o", 1e-6, 1e-5)
    check(rays_d, rd_c, "rays_d", 1e-6, 1e-4, floor=1e-2)
    check(ret["rgb"], ref["rgb"], "rendered colour")
    check(ret["depth"], ref["depth"], "rendered depth")
    for k in ("rgb_loss", "depth_loss", "sdf_loss", "fs_loss", "psnr"):
        check(ret[k], ref[k], k, 1e-4, 1e-4, floor=1e-6)
    check(loss, ref_loss, "total loss", 1e-4, 1e-4, floor=1e-6)
    check_grad(m.embed_fn.params.grad, cpu.embed_fn.params.grad, "grid gradient (9.0 M entries)")
    for (k, p), (_, q) in zip(m.decoder.named_parameters(), cpu.decoder.named_parameters()):
        check_grad(p.grad, q.grad, "d " + k)
    check_grad(rot.grad, rot_c.grad, "d quaternions", 5e-4, 0.0)
    check_grad(trans.grad, trans_c.grad, "d translations", 5e-4, 0.0)
    # ---- one dense Adam step on both sides (mipsfusion.py:580-584)
    FusedAdam(map_groups(m, cfg), betas=(0.9, 0.99)).step()
    torch.optim.Adam(map_groups(cpu, cfg), betas=(0.9, 0.99)).step()
    # first Adam step = -lr * sign(g) wherever |g| >> eps: entries whose gradient is round-off around zero may flip.
    # "Solid" = above the rounding noise of BOTH sides: the oracle is fp32 itself and sits ~1e-5 of the maximum away from
    # the fp64 truth (tools/dbg_bwd_accuracy.py; the f16x3 kernels are closer to the truth than the fp32-MFMA ones, but
    # their error is independent of the oracle's instead of correlated with it)
    a, b = _np(m.embed_fn.params), _np(cpu.embed_fn.params)
    g = np.abs(_np(cpu.embed_fn.params.grad))
    solid = g > 1e-3 * g.max()
    print(f"  grid after Adam: {int(solid.sum())} entries with a solid gradient, max |diff| there "
          f"{np.abs(a - b)[solid].max():.2e}; untouched entries equal: {np.array_equal(a[g == 0], b[g == 0])}")
    assert np.abs(a - b)[solid].max() < 1e-4 * 0.01 + 1e-7, "grid entries after one Adam step"
    assert np.array_equal(a[g == 0], b[g == 0]), "entries with zero gradient must not move differently"
    for (k, p), (_, q) in zip(m.decoder.named_parameters(), cpu.decoder.named_parameters()):
        check_grad(p, q, "after Adam: " + k, 1e-4, 1e-3)


def test_config2_eval_forward_vs_oracle_full(dev):
    """Forward-only (eval) at full size: raw network output, z_vals bit-exact, rendered maps."""
    cfg = synth.config_headline()
    m, cpu = build(cfg, dev, seed=3)
    m.eval()
    f = synth.make_frame(cfg, seed=4)
    H, W = f["depth"].shape
    random.seed(4), torch.manual_seed(4)
    idx = torch.tensor(random.sample(range(H * W), 4096))
    ro, rd, rgb, d = synth.ray_batch(f, idx, f["c2w"])
    noise = torch.rand(4096, 64)
    with torch.no_grad():
        out = m.forward(ro.to(dev), rd.to(dev), None, d.to(dev), noise=noise.to(dev))
        ref = cpu.render_rays(ro, rd, d, noise)
    assert np.array_equal(out["z_vals"].cpu().numpy(), ref["z_vals"].numpy()), "sample placement must be bit-exact"
    print("\nconfig 2, eval forward, 4096 x 64, hash 2^19:")
    # sdf = (sum_c p_c * c / 4 - 0.5) * 2 (decoder.py:72) is a difference of O(1) terms: fp32 rounding of those terms
    # alone is ~1e-7 ABSOLUTE, so a per-element relative bound of 1e-4 is only meaningful above |sdf| ~ 3e-3; the floor
    # is 1e-2 (= 1 mm at trunc 0.1 m)
    check(out["raw"][..., 3], ref["raw"][..., 3], "sdf (262 144 samples)", floor=1e-2)
    check(out["raw"][..., :3], ref["raw"][..., :3], "raw colour", floor=1e-2)
    check(out["raw"][..., 5:], ref["raw"][..., 5:], "class probabilities", floor=1e-2)
    check(out["rgb"], ref["rgb"], "rendered colour")
    check(out["depth"], ref["depth"], "rendered depth")
    check(out["depth_var"], ref["depth_var"], "depth variance", 1e-4, 1e-3, floor=1e-3)


# ------------------------------------------------------------------------------------------------ config 5
def test_config5_scannet_tracking_iteration_vs_oracle(dev):
    """ScanNet workload: one gradient-tracking iteration exactly as tracking_render does it with iter_RO = 0
    (mipsfusion.py:508-534: select_samples over the edge-cropped image, the H-for-W quirk of :514 included),
    1000 rays x 75 samples (more than one wavefront per ray), far 7, pose gradient through the fused ray op."""
    cfg = synth.config_scannet()
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    assert (fx, fy, cx, cy) == (577.0, 578.0, 308.0, 232.0), "floor-division quirk of datasets/dataset.py:29-30"
    m, cpu = build(cfg, dev, seed=5)
    m.train()
    f = synth.make_frame(cfg, seed=5)
    iH, iW = cfg["tracking"]["ignore_edge_H"], cfg["tracking"]["ignore_edge_W"]
    n = cfg["tracking"]["sample"]
    random.seed(5), torch.manual_seed(5)
    indice = sh.select_samples(H - iH * 2, W - iW * 2, n)
    ih, iw = torch.remainder(indice, H - iH * 2), torch.div(indice, H - iH * 2, rounding_mode="floor")
    d_cam = f["direction"][iH:-iH, iW:-iW, :][ih, iw, :]
    t_s, t_d = f["rgb"][iH:-iH, iW:-iW, :][ih, iw, :], f["depth"][iH:-iH, iW:-iW][ih, iw].unsqueeze(-1)
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    assert S == 75
    noise = torch.rand(n, S)
    pose = f["c2w"].clone()
    pose[:3, 3] += torch.tensor([0.01, -0.02, 0.015])
    # product
    rot = torch.nn.Parameter(matrix_to_quaternion(pose[None, :3, :3]).to(dev))
    trans = torch.nn.Parameter(pose[None, :3, 3].clone().to(dev))
    own = torch.zeros(n, dtype=torch.int64, device=dev)
    rays_o, rays_d = ops.pose_rays(rot, trans, None, own, d_cam.contiguous().to(dev))
    ret = m.forward(rays_o, rays_d, t_s.contiguous().to(dev), t_d.contiguous().to(dev), EMD_w=0., noise=noise.to(dev))
    loss = path_cpu.total_loss(ret, cfg["training"])
    loss.backward()
    # oracle
    rot_c = torch.nn.Parameter(matrix_to_quaternion(pose[None, :3, :3]))
    trans_c = torch.nn.Parameter(pose[None, :3, 3].clone())
    c2w = qt_to_transform_matrix(rot_c, trans_c)
    ro_c = c2w[..., :3, -1].repeat(n, 1)
    rd_c = torch.sum(d_cam[..., None, :] * c2w[:, :3, :3], -1)
    ref = cpu.train_forward(ro_c, rd_c, t_s, t_d, noise, 0.0)
    ref_loss = path_cpu.total_loss(ref, cfg["training"])
    ref_loss.backward()
    print("\nconfig 5 (ScanNet bound / intrinsics / far 7), tracking iteration 1000 x 75:")
    check(ret["rgb"], ref["rgb"], "rendered colour")
    check(ret["depth"], ref["depth"], "rendered depth")
    for k in ("rgb_loss", "sdf_loss", "fs_loss"):
        check(ret[k], ref[k], k, 1e-4, 1e-4, floor=1e-6)
    check(loss, ref_loss, "total loss", 1e-4, 1e-4, floor=1e-6)
    check_grad(rot.grad, rot_c.grad, "d quaternion", 5e-4, 0.0)
    check_grad(trans.grad, trans_c.grad, "d translation", 5e-4, 0.0)
    check_grad(m.embed_fn.params.grad, cpu.embed_fn.params.grad, "grid gradient (computed, then discarded)")


def _scannet_ro(cfg, dev, frame):
    from mipsfusion_amd.RandomOptimizer import RandomOptimizer
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frame["direction"])
    np.random.seed(7)
    return RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))


def test_config5_scannet_random_optimizer_slice_vs_oracle(dev):
    """The forward-only slice BASELINE config 5 names: 2000 particles x (16 x 24) lattice points through
    normalisation -> hash grid 2^19 -> SDF decoder -> masked mean |sdf| at the ScanNet bound, against the oracle's
    network on the same 768 000 points; then three full rounds against oracle/ro_cpu.py driven by the ORACLE network."""
    cfg = synth.config_scannet()
    m, cpu = build(cfg, dev, seed=6)
    m.eval()
    f = synth.make_frame(cfg, seed=6)
    ro = _scannet_ro(cfg, dev, f)
    assert ro.decoder_precision == "bf16x6"        # the default: the reference's fp32 arithmetic; the opt-in plain f16 is checked at the end
    init = f["c2w"].clone()
    init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
    P, n = ro.particle_size, ro.row_indices.shape[0]
    assert (P, n) == (2000, 384)
    # ---- one fitness evaluation, particle by particle
    state = torch.zeros(ops.RO_STATE_FLOATS, device=dev)
    state[0:9], state[9:12], state[12:18] = init[:3, :3].reshape(9).to(dev), init[:3, 3].to(dev), 0.02
    td = f["depth"][ro.row_indices, ro.col_indices].to(dev).contiguous()
    with torch.no_grad():
        mm = ro._enqueue_round(m, state.clone(), td, ro._dirs[0], m._rc(1, 0),
                               ops.decoder_pack16(m.decoder.ordered_parameters(), precision=ro.decoder_precision))
        pst7 = ro_cpu.pose_6d_to_7d(ro.pre_sampled_particle.cpu() * 0.02)
        cam = f["direction"][ro.row_indices, ro.col_indices, :] * td.cpu()[:, None]
        world, _, _ = ro_cpu.particle_points(init[:3, :3], init[:3, 3:], pst7, cam)
        mm_ref = ro_cpu.mean_masked_sdf(cpu.run_network, world, td.cpu()[:, None], cfg["training"]["trunc"])
    print("\nconfig 5 (ScanNet), RandomOptimizer slice 2000 x 384:")
    check(mm, mm_ref, "mean masked |sdf| per particle", 1e-4, 1e-4, floor=1e-5)
    # ---- three rounds, oracle end to end
    pose, st = ro.optimize(m, f["depth"], init, None, n_iter=3, return_state=True)
    with torch.no_grad():
        ref_pose, trace = ro_cpu.optimize(cpu.run_network, ro.pre_sampled_particle.cpu(), ro.row_indices,
                                          ro.col_indices, f["depth"], f["direction"], init, 3, 0.02, 0.5,
                                          cfg["training"]["trunc"])
    check(pose[:3, 3], ref_pose[:3, 3], "tracked translation after 3 rounds", 1e-4, 1e-4)
    check(pose[:3, :3], ref_pose[:3, :3], "tracked rotation after 3 rounds", 1e-4, 1e-3, floor=1e-2)
    check(st[12:18], trace[-1]["search"].reshape(6), "search size", 1e-3, 1e-3, floor=1e-6)
    # ---- BASELINE config 5's "fp16 decoder on CDNA4": plain f16 matrix-core operands (opt-in)
    ro.decoder_precision = "f16"
    with torch.no_grad():
        mm16 = ro._enqueue_round(m, state.clone(), td, ro._dirs[0], m._rc(1, 0),
                                 ops.decoder_pack16(m.decoder.ordered_parameters(), precision=ro.decoder_precision))
    pose16 = ro.optimize(m, f["depth"], init, None, n_iter=3)
    e_fit = float((mm16.cpu() - mm_ref).abs().max() / mm_ref.abs().max())
    e_pose = float((pose16.cpu() - ref_pose).abs().max())
    print(f"  plain f16 decoder: fitness error {e_fit:.2e} of max, tracked pose error {e_pose:.2e} (tolerances 5e-3, 1e-3)")
    assert e_fit < 5e-3 and e_pose < 1e-3


# ------------------------------------------------------------------------------------------------ config 4
def test_config4_global_ba_single_rank_vs_oracle(dev):
    """Cross-sub-map global BA (InactiveMap.py:375-474) over three FastCaMo-large-style sub-maps (hash 2^16,
    centre-length normalisation) on one rank: anchors and loss trace vs oracle/global_ba_cpu.py with CPU copies of the
    same networks.  The sharded form of the same class is covered with gloo world-2 in tests/test_dist_cpu.py and with
    two processes on this GPU in test_two_process_run_exercises_every_sharding."""
    from mipsfusion_amd.global_ba import PairTerm, ShardedGlobalBA, frozen, model_query
    from oracle import global_ba_cpu
    cfg = synth.config_large_submap()
    models, cpus = [], []
    for s in range(3):
        m, c = build(cfg, dev, seed=20 + s, grid_scale=0.3)
        models.append(m.eval()), cpus.append(c)
    anchors = torch.eye(4)[None].repeat(3, 1, 1)
    for s in (1, 2):
        anchors[s] = synth.default_pose(cfg, yaw=0.08 * s, pitch=0.02 * s)
        anchors[s, :3, 3] = torch.tensor([0.3 * s, -0.2 * s, 0.05 * s])
    f = synth.make_frame(cfg, seed=9)
    f7 = torch.cat([f["direction"], f["rgb"], f["depth"][..., None]], -1).reshape(-1, 7)
    g = torch.Generator().manual_seed(5)
    bs, n_iter = 900, 6
    batches = []
    for _ in range(n_iter):
        terms = []
        for (i, j) in ((0, 1), (1, 2)):
            idx = torch.randint(0, f7.shape[0], (bs,), generator=g)
            kf = f["c2w"][None].repeat(bs, 1, 1).clone()
            kf[:, :3, 3] += 0.02 * torch.randn(bs, 3, generator=g)
            terms.append((i, j, f7[idx], kf, 5.0, None))
        idx = torch.randint(0, f7.shape[0], (400,), generator=g)
        terms.append((2, 0, f7[idx], f["c2w"][None].clone(), 100.0, (torch.rand(400, 1, generator=g) > 0.3).float()))
        batches.append(terms)
    trunc = cfg["training"]["trunc"]
    ref_poses, ref_trace = global_ba_cpu.optimise(cpus, anchors, batches, trunc, pose_accum_step=2)
    with frozen(models):
        ba = ShardedGlobalBA(model_query(dict(enumerate(models))), range(3), anchors.to(dev), trunc, pose_accum_step=2)
        trace = [float(ba.iteration([PairTerm(i, j, r.to(dev), k.to(dev), w, None if mk is None else mk.to(dev))
                                     for (i, j, r, k, w, mk) in terms])) for terms in batches]
    assert all(p.grad is None for m in models for p in m.parameters()), "the networks are frozen during global BA"
    print("\nconfig 4, global BA over 3 sub-maps (hash 2^16, centre-length normalisation):")
    check(torch.tensor(trace), torch.tensor(ref_trace), "loss trace (6 iterations)", 1e-3, 1e-3, floor=1e-6)
    moved = (ref_poses[1:] - anchors[1:]).abs().max()
    assert moved > 1e-3, "the anchors must move"
    err = (ba.result().cpu() - ref_poses).abs().max()
    print(f"  anchors moved by up to {moved:.2e}; max |product - oracle| {err:.2e}")
    assert err < 2e-5 + 1e-2 * moved


def _pose_only_tracking_iteration(dev, seed=9):
    """-> (cfg, model, frame, fn): fn() runs one pose-only tracking iteration (frozen map, fixed rays and jitter) and returns its
    pose gradients and depth map -- deterministic by construction (no float atomics anywhere on that path)."""
    cfg = synth.config_headline()
    m, _ = build(cfg, dev, seed=seed)
    m.train()
    for p in m.parameters():
        p.requires_grad_(False)                                      # tracking: the map is frozen (mipsfusion.py:226-230)
    f = synth.make_frame(cfg, seed=seed)
    H, W = f["depth"].shape
    g = torch.Generator().manual_seed(3)
    idx = torch.randperm(H * W, generator=g)[:cfg["tracking"]["sample"]]
    r, c = torch.div(idx, W, rounding_mode="floor"), torch.remainder(idx, W)
    d_cam, rgb, d = f["direction"][r, c].to(dev), f["rgb"][r, c].contiguous().to(dev), f["depth"][r, c][:, None].contiguous().to(dev)
    noise = torch.rand(idx.numel(), 64, generator=g).to(dev)
    q0, t0 = matrix_to_quaternion(f["c2w"][None, :3, :3]).to(dev), f["c2w"][None, :3, 3].clone().to(dev)
    owner = torch.zeros(idx.numel(), dtype=torch.int64, device=dev)

    def tracking_iteration():
        rot, trans = torch.nn.Parameter(q0.clone()), torch.nn.Parameter(t0.clone())
        ro, rd = ops.pose_rays(rot, trans, None, owner, d_cam)
        ret = m.forward(ro, rd, rgb, d, EMD_w=0.0, noise=noise)
        path_cpu.total_loss(ret, cfg["training"]).backward()
        return torch.cat([rot.grad.reshape(-1), trans.grad.reshape(-1), ret["depth"].detach().reshape(-1)])
    return cfg, m, f, tracking_iteration


# ------------------------------------------------------------------------ the reference's process topology: two processes, one GPU
def test_mapping_and_tracking_beside_a_second_process_on_the_same_gpu(dev):
    """The reference runs TWO processes on one GPU: the active map's tracking + mapping and the InactiveMap process's local BA
    (mipsfusion.py:661-667, shared modules :107-124, InactiveMap.py:203-308).  Here process B (tools/ba_load.py) runs mapping
    steps back to back on this device while THIS process takes (a) the complete config-2 mapping iteration and (b) the config-5
    tracking iteration against the oracle at their normal gates, and (c) repeats a pose-only tracking iteration and a
    RandomOptimizer frame 200 times each: everything on that path is deterministic (no float atomics: fixed-order reductions,
    fp64 LDS accumulation rounded once), so every repetition must reproduce the first BIT FOR BIT -- a wavefront disturbed by
    the neighbour process shows here.  It did: DESIGN.md 4h -- with the decoder kernels of the neighbour on the same CUs, a
    packed fp32 add whose low result takes the high half of a source lost that operand in lanes 48..63 (28 of 200 RandomOptimizer
    frames on the build of that day); the kernels that would hold such an instruction are now built without packed fp32
    (MIPSF_SINGLE_FP32, tools/audit_packed.py) and this test is what watches the rest."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    other = subprocess.Popen([sys.executable, os.path.join(root, "tools", "ba_load.py"), "--seconds", "240"], cwd=root, env=env,
                             stdout=subprocess.PIPE, text=True)
    try:
        line = other.stdout.readline()
        assert line.strip() == "READY", f"the second process did not start: {line!r}"
        test_config2_full_iteration_vs_oracle(dev)                       # (a)
        test_config5_scannet_tracking_iteration_vs_oracle(dev)           # (b)
        # (c) run-to-run identity under sharing
        cfg, m, f, tracking_iteration = _pose_only_tracking_iteration(dev)
        first = tracking_iteration()
        n_diff = sum(0 if torch.equal(tracking_iteration(), first) else 1 for _ in range(200))
        print(f"\n  beside a second process: {n_diff} of 200 tracking iterations differ from the first")
        assert n_diff == 0
        from mipsfusion_amd.RandomOptimizer import RandomOptimizer
        cfg["tracking"]["RO"].update(initial_scaling_factor=0.02, rescaling_factor=0.5)
        cfg["tracking"]["ignore_edge_W"] = cfg["tracking"]["ignore_edge_H"] = 20
        Hc, Wc, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
        ds = types.SimpleNamespace(H=Hc, W=Wc, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=f["direction"])
        np.random.seed(5)
        ro_ = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
        m.eval()
        init = f["c2w"].clone()
        init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
        p_first = ro_.optimize(m, f["depth"], init, None, n_iter=5).clone()
        n_diff = sum(0 if torch.equal(ro_.optimize(m, f["depth"], init, None, n_iter=5), p_first) else 1 for _ in range(200))
        print(f"  beside a second process: {n_diff} of 200 RandomOptimizer frames (5 rounds each) differ from the first")
        assert n_diff == 0
        assert other.poll() is None, "the second process ended before the checks did"
    finally:
        other.terminate()
        try:
            other.wait(30)
        except subprocess.TimeoutExpired:
            other.kill()


def test_deterministic_kernels_beside_the_decoder_kernels_on_another_stream(dev):
    """DESIGN.md 4h, the setting in which the packed-fp32 hazard showed best (3 % of the launches of the build of that day): the
    decoder's forward and backward kernels taking turns on one stream, the RandomOptimizer's particle kernel -- and then whole
    pose-only tracking iterations -- launched over and over on another one.  Every launch must reproduce the first bit for bit."""
    import time
    cfg = synth.config_headline()
    m, _ = build(cfg, dev, seed=11)
    m.train()
    M = 4096 * 64
    g = torch.Generator().manual_seed(2)
    xn = torch.rand(M, 3, generator=g).to(dev)
    ws = m.decoder.ordered_parameters()
    from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR
    feat = ops.hashgrid_fwd(xn, m.embed_fn.params.detach(), m.embed_fn.meta, FEAT_LEVEL_MAJOR)
    dout = (torch.randn(M, 10, generator=g) * 1e-3).to(dev)
    pk = ops.decoder_pack16(ws, precision=m.decoder_precision)

    def decoder_round():                                     # forward (masks only) then the backward chain, as in a tracking step
        out, saved = ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, xn, None, M, "masks", precision=m.decoder_precision, packed16=pk)
        ops.decoder_bwd(None, feat, FEAT_LEVEL_MAJOR, xn, None, out, dout, saved, None, M, precision=m.decoder_precision, packed16=pk)
    decoder_round()
    P, n = 2000, 384
    pst = (torch.rand(P, 6, generator=g) * 2 - 1).to(dev)
    state = torch.zeros(ops.RO_STATE_FLOATS, device=dev)
    state[:12] = torch.tensor([0.962, -0.059, 0.266, 0.011, 0.984, 0.178, -0.272, -0.169, 0.947, 1.168, 3.796, 0.946])
    state[12:18] = 0.02
    dirs = torch.stack([torch.rand(n, generator=g) - 0.5, 0.8 * (torch.rand(n, generator=g) - 0.5), torch.ones(n)], 1).contiguous().to(dev)
    depth = (0.8 + 2.2 * torch.rand(n, generator=g)).to(dev)
    rc = m._rc(1, 0)
    side = torch.cuda.Stream(dev)
    with torch.cuda.stream(side):
        ref, ref7 = ops.ro_particles(pst, state, dirs, depth, rc, point_major=True)
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    launches, t0 = 0, time.time()
    while time.time() - t0 < 6.0:
        for _ in range(4):
            decoder_round()
        with torch.cuda.stream(side):
            for _ in range(32):
                a, b = ops.ro_particles(pst, state, dirs, depth, rc, point_major=True)
                bad += (a != ref).any() | (b != ref7).any()
                launches += 1
        torch.cuda.synchronize()
    print(f"\n  {int(bad)} of {launches} particle-kernel launches beside the decoder kernels differ from the first")
    assert launches > 2000 and int(bad) == 0
    # the tracking iteration's kernels (pose rays, placement, render forward / backward, their pose gradients) the same way
    _cfg, _m, _f, tracking_iteration = _pose_only_tracking_iteration(dev)
    with torch.cuda.stream(side):
        first = tracking_iteration()
    torch.cuda.synchronize()
    n_diff, n_it, t0 = 0, 0, time.time()
    while time.time() - t0 < 6.0:
        for _ in range(4):
            decoder_round()
        with torch.cuda.stream(side):
            for _ in range(4):
                n_diff += 0 if torch.equal(tracking_iteration(), first) else 1
                n_it += 1
        torch.cuda.synchronize()
    print(f"  {n_diff} of {n_it} tracking iterations beside the decoder kernels differ from the first")
    assert n_it > 100 and n_diff == 0


def test_particle_kernel_beside_back_to_back_mfmas(dev, tmp_path):
    """The strongest known trigger of the packed-fp32 hazard: tools/micro/pk_lanes.hip launches a kernel whose
    waves issue bf16 MFMAs back to back on a second stream and, beside it, (a) its own victim kernel with hipcc's crossed packed
    add -- reported, not asserted: ~6 % of the launches come out wrong on the boxes seen so far -- and (b) the LIBRARY's particle
    kernel on the same inputs, which must reproduce its first launch every time."""
    import os
    import re
    import shutil
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not shutil.which(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "pk_lanes")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-w", "-I" + os.path.join(root, "include"), "-o", exe,
                    os.path.join(root, "tools", "micro", "pk_lanes.hip"), "-ldl"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    from mipsfusion_amd import _lib
    counts = {}
    for mode in ("asm", "lib:" + _lib.LIB_PATH):
        out = subprocess.run([exe, mode, "5", "inproc:10"], check=True, capture_output=True, text=True, timeout=120).stdout
        m = re.search(r"(\d+) of (\d+) launches differ", out)
        assert m, out
        counts[mode[:3]] = (int(m.group(1)), int(m.group(2)))
    print(f"\n  beside back-to-back bf16 MFMAs: crossed packed add {counts['asm'][0]} of {counts['asm'][1]} launches wrong; "
          f"the library's particle kernel {counts['lib'][0]} of {counts['lib'][1]}")
    assert counts["lib"][1] > 5000 and counts["lib"][0] == 0


# ------------------------------------------------------------------------ ray-data-parallel training == the single-process step
RDP_STEPS, RDP_N, RDP_S = 5, 4096, 64


def _rdp_batches(cfg):
    """five batches of 4096 rays of one synthetic frame + their jitter: the same on every rank, in the oracle and in the parent"""
    f = synth.make_frame(cfg, seed=11)
    H, W = f["depth"].shape
    g = torch.Generator().manual_seed(123)
    out = []
    for _ in range(RDP_STEPS):
        idx = torch.randperm(H * W, generator=g)[:RDP_N]
        ro, rd, rgb, d = synth.ray_batch(f, idx, f["c2w"])
        out.append((ro, rd, rgb, d, torch.rand(RDP_N, RDP_S, generator=g)))
    return out


def _rdp_gpu_worker(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)        # (both ranks sit on the one GPU of the box)
    try:
        from mipsfusion_amd.ray_dp import RayDataParallelStep
        dev = torch.device("cuda:0")
        cfg = synth.config_headline()
        m, _ = build(cfg, dev, seed=21)
        m.train()
        mp_ = cfg["mapping"]
        rdp = RayDataParallelStep(
            m, lambda shard: FusedAdam([{"params": [shard], "eps": 1e-15, "lr": mp_["lr_embed"]}], betas=(0.9, 0.99)),
            lambda ps: FusedAdam([{"params": ps, "weight_decay": 1e-6, "lr": mp_["lr_decoder"]}], betas=(0.9, 0.99)))
        b, e = rdp.my_share(RDP_N)
        losses, first_grads = [], None
        for ro, rd, rgb, d, noise in _rdp_batches(cfg):
            ret = m.forward(ro[b:e].to(dev), rd[b:e].to(dev), rgb[b:e].contiguous().to(dev), d[b:e].contiguous().to(dev),
                            EMD_w=0.01, noise=noise[b:e].to(dev))
            loss = path_cpu.total_loss(ret, cfg["training"])
            loss.backward()
            if first_grads is None:         # this rank's part of the first step's gradient (the parent adds the two parts)
                first_grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
            rdp.step()
            losses.append([float(loss.detach())] + [float(ret[k].detach()) for k in ("rgb_loss", "depth_loss", "sdf_loss", "fs_loss")])
        torch.save({"state": {k: v.cpu() for k, v in m.state_dict().items()}, "losses": losses, "first_grads": first_grads},
                   os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ray_data_parallel_two_ranks_equal_the_single_process_step_and_the_oracle(dev):
    """SURVEY 8e row 2 on the real JointEncoding at BASELINE config 2: two processes (here both on this GPU, gloo) render
    2048 rays each of the SAME 4096-ray batches for five steps -- the batch's nine loss sums all-reduced inside forward
    (helper_functions/utils.py:43-47: fs_weight / sdf_weight from counts over the WHOLE batch; scene_rep.py:218: depth_loss over
    the batch's valid rays), gradients summed, reduce-scatter -> Adam on half the table each -> all-gather -- against ONE
    process that takes the same five steps on the whole batches (mipsfusion.py:325-335), and against the oracle's
    train_forward + torch.optim.Adam.  Both ranks must hold bit-identical parameters; the losses of every step must be the
    whole batch's on both."""
    import socket
    import tempfile
    import torch.multiprocessing as tmp
    cfg = synth.config_headline()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    with tempfile.TemporaryDirectory() as out_dir:
        ctx = tmp.get_context("spawn")
        procs = [ctx.Process(target=_rdp_gpu_worker, args=(r, 2, port, out_dir)) for r in range(2)]
        for p in procs:
            p.start()
        import time as _time
        deadline = _time.time() + 420
        while _time.time() < deadline and any(p.is_alive() for p in procs) and all(p.exitcode in (None, 0) for p in procs):
            _time.sleep(0.5)
        codes = [p.exitcode for p in procs]
        for p in procs:                      # (a rank that died leaves the other waiting in a collective: do not wait for it)
            if p.is_alive():
                p.terminate()
            p.join(30)
        assert codes == [0, 0], f"ray-DP ranks ended with exit codes {codes}"
        r0, r1 = (torch.load(f"{out_dir}/rank{r}.pt") for r in range(2))
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), f"ranks diverged: {k}"
    assert r0["losses"] == r1["losses"], "every rank must report the whole batch's losses"
    # ---- one process, whole batches
    m, cpu = build(cfg, dev, seed=21)
    m.train()
    mp_ = cfg["mapping"]
    opt = FusedAdam(map_groups(m, cfg), betas=(0.9, 0.99))
    copt = torch.optim.Adam(map_groups(cpu, cfg), betas=(0.9, 0.99))
    one_losses, ref_losses, one_first = [], [], None
    for ro, rd, rgb, d, noise in _rdp_batches(cfg):
        ret = m.forward(ro.to(dev), rd.to(dev), rgb.to(dev), d.to(dev), EMD_w=0.01, noise=noise.to(dev))
        loss = path_cpu.total_loss(ret, cfg["training"])
        loss.backward()
        if one_first is None:
            one_first = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
        opt.step()
        opt.zero_grad()
        one_losses.append([float(loss.detach())] + [float(ret[k].detach()) for k in ("rgb_loss", "depth_loss", "sdf_loss", "fs_loss")])
        ref = cpu.train_forward(ro, rd, rgb, d, noise, 0.01)
        rl = path_cpu.total_loss(ref, cfg["training"])
        copt.zero_grad()
        rl.backward()
        copt.step()
        ref_losses.append([float(rl)] + [float(ref[k]) for k in ("rgb_loss", "depth_loss", "sdf_loss", "fs_loss")])
    print("\nray-DP (2 ranks x 2048 rays) vs one process vs oracle, 5 steps at config 2:")
    lo_dp, lo_one, lo_ref = (np.array(x) for x in (r0["losses"], one_losses, ref_losses))
    print(f"  losses: worst |dp - one| / one {np.abs(lo_dp / lo_one - 1).max():.2e}, worst |dp - oracle| / oracle "
          f"{np.abs(lo_dp / lo_ref - 1).max():.2e}")
    assert np.abs(lo_dp / lo_one - 1).max() < 2e-5 and np.abs(lo_dp / lo_ref - 1).max() < 1e-4

    def compare(a, b, what, tol, frac):
        a, b = _np(a).ravel(), _np(b).ravel()
        scale = np.abs(b).max() + 1e-30
        off = np.abs(a - b) > tol * scale
        print(f"  {what:44s} max |diff| / max {np.abs(a - b).max() / scale:.2e}, entries off by > {tol:g} of max: "
              f"{int(off.sum())} of {off.size}")
        assert off.mean() <= frac, f"{what}: {int(off.sum())} entries off"
    # ---- the GRADIENT of the first step: the two ranks' parts added = the single process's, to the order of fp32 additions
    #      (EVERY entry within 1e-5 of the tensor's maximum)
    for k, g_one in one_first.items():
        g_dp = r0["first_grads"][k] + r1["first_grads"][k]
        compare(g_dp, g_one, "first-step gradient, dp sum vs one process: " + k, 1e-5, 0.0)
    one = {k: v.cpu() for k, v in m.state_dict().items()}
    ora = {k: v for k, v in cpu.state_dict().items()}
    for k in r0["state"]:
        if r0["state"][k].numel() == 0:
            continue
        # ---- the PARAMETERS after five Adam steps.  Adam turns the SIGN of a gradient entry into a step of lr: a table entry
        # whose gradient is rounding noise around zero (the far corner of a cell a single sample grazed) moves by +-0.01 per step
        # in ANY two fp32 evaluations that add in a different order -- 2e-4 of the 9 M entries here; every other entry must
        # agree to 1e-5 of the tensor's maximum, and the update as a whole to 1 % in the L2 norm.  Against the fp32 oracle the
        # usual gates.
        # (the decoder's 36 577 weights all have solid gradients, but Adam divides by sqrt(v): a weight whose gradient is 1e-3 of
        # its tensor's largest carries the 1e-6 relative reordering noise of that largest one as 1e-3 of its own step)
        grid = k.startswith("embed_fn")
        compare(r0["state"][k], one[k], "dp vs one process: " + k, 1e-5 if grid else 2e-4, 1e-3)
        compare(r0["state"][k], ora[k], "dp vs oracle + torch Adam: " + k, 1e-3, 2e-3)
    init, _ = build(cfg, dev, seed=21)
    p0 = init.embed_fn.params.detach().cpu()
    upd_dp, upd_one = r0["state"]["embed_fn.params"] - p0, one["embed_fn.params"] - p0
    rel = float((upd_dp - upd_one).norm() / upd_one.norm())
    print(f"  table update over the five steps: |dp - one| / |one| = {rel:.2e} (L2)")
    assert rel < 1e-2


def test_two_process_run_exercises_every_sharding(dev):
    """`python bench.py --gpus 2` WITHOUT a launcher: bench.py starts its two ranks itself (child processes of
    torch.distributed.run, before it touches the GPU) and passes rank 0's line through; here both ranks sit on this one GPU
    over gloo (RCCL needs one GPU per rank): sub-map-per-rank mapping steps + pose all_gather, the RandomOptimizer particle
    split and the global-BA pose-gradient all-reduce (bench.multi_gpu_checks)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIPSF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
           "--warmup", "5", "--setup-iters", "10", "--cpu-rays", "0", "--no-frame-estimate"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(line) < 6000                      # the line the driver parses (bench.compact_line); the full result is beside it
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["multi_gpu"]["ranks"]["world_size"] == 2 and len(out["multi_gpu"]["ms_per_step_of_each_rank"]) == 2
    detail = json.load(open(os.path.join(root, "bench_detail.json")))
    assert detail["value"] == out["value"]
    mg = detail["multi_gpu"]
    assert mg["ranks"]["world_size"] == 2 and mg["ranks"]["backend"] == "gloo" and len(mg["ranks"]["device_of_rank"]) == 2
    print("\ntwo ranks on one GPU (gloo):", json.dumps(mg))
    assert mg["ro_split_pose_equals_unsplit"] is True
    assert mg["ro_particles_per_rank"] == 1000
    assert mg["global_ba_anchor_spread_over_ranks"] == 0.0 and mg["global_ba_anchors_moved"] is True
    # ray-data-parallel training: both ranks render half the batch, reduce-scatter / sharded Adam / all-gather
    rdp = mg["ray_dp_training"]
    assert rdp["params_equal_over_ranks"] is True and rdp["rays_per_rank"] == 2048
    assert rdp["loss_first_last"][1] < rdp["loss_first_last"][0]
    assert mg["pose_all_gather_ms"] > 0 and mg["pose_grad_all_reduce_ms"] > 0 and len(mg["ms_per_step_of_each_rank"]) == 2
