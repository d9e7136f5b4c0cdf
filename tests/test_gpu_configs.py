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
    assert ro.decoder_precision == "f16x3"         # the default: parity arithmetic; the opt-in plain f16 is checked at the end
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
                               ops.decoder_pack16(m.decoder.ordered_parameters()))
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
                                 ops.decoder_pack16(m.decoder.ordered_parameters()))
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


def test_two_process_run_exercises_every_sharding(dev):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), here with both
    ranks on this one GPU and the gloo backend (RCCL needs one GPU per rank): sub-map-per-rank mapping steps + pose
    all_gather, the RandomOptimizer particle split and the global-BA pose-gradient all-reduce (bench.multi_gpu_checks)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MIPSF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
           "--warmup", "5", "--setup-iters", "10", "--cpu-rays", "0", "--no-frame-estimate"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    mg = out["multi_gpu"]
    print("\ntwo ranks on one GPU (gloo):", json.dumps(mg))
    assert mg["ro_split_pose_equals_unsplit"] is True
    assert mg["ro_particles_per_rank"] == 1000
    assert mg["global_ba_anchor_spread_over_ranks"] == 0.0 and mg["global_ba_anchors_moved"] is True
    # ray-data-parallel training: both ranks render half the batch, reduce-scatter / sharded Adam / all-gather
    rdp = mg["ray_dp_training"]
    assert rdp["params_equal_over_ranks"] is True and rdp["rays_per_rank"] == 2048
    assert rdp["loss_first_last"][1] < rdp["loss_first_last"][0]
    assert mg["pose_all_gather_ms"] > 0 and mg["pose_grad_all_reduce_ms"] > 0 and len(mg["ms_per_step_of_each_rank"]) == 2
