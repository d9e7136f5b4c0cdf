"""CPU: pin the oracle (oracle/path_cpu.py, oracle/p3d_cpu.py) against the golden vectors the
REFERENCE produced (tests/golden/make_golden.py), and -- when /root/reference is present, i.e. in the
build container -- against the reference imported live."""
import math

import numpy as np
import pytest
import torch

from oracle import p3d_cpu, path_cpu, ref_import, tcnn_cpu
from mipsfusion_amd import synth

from .conftest import load_golden


def T(a):
    return torch.from_numpy(np.asarray(a))


def _np(a):
    if torch.is_tensor(a):
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = _np(a), _np(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def cfg_for(name):
    cfg = synth.config_plumbing()
    if name == "scene_s75.npz":
        cfg["training"].update(n_samples_d=50, n_range_d=25, n_samples=75)
    if name == "scene_nd0.npz":         # scene_rep.py:166-167: no uniform samples, z_vals = the depth-guided ones
        cfg["training"].update(n_samples_d=0, n_range_d=16, n_samples=16)
    return cfg


def scene_from(g, cfg):
    m = path_cpu.CpuScene(cfg, g["bound"], g["half_len"])
    sd = {k[2:]: T(g[k]) for k in g.files if k.startswith("w.")}
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("name", ["scene_cfg1.npz", "scene_s75.npz", "scene_nd0.npz"])
def test_scene_eval_matches_reference(name):
    g = load_golden(name)
    cfg = cfg_for(name)
    m = scene_from(g, cfg)
    with torch.no_grad():
        out = m.render_rays(T(g["rays_o"]), T(g["rays_d"]), T(g["target_d"]), T(g["noise"]))
    assert np.array_equal(out["z_vals"].numpy(), g["eval.z_vals"]), "sample placement must be bit-exact"
    for k in ("raw", "rgb", "depth", "disp_map", "acc_map", "depth_var"):
        close(out[k], g["eval." + k], rtol=2e-5, atol=2e-6)
    with torch.no_grad():
        nd = m.render_rays(T(g["rays_o"]), T(g["rays_d"]), None, T(g["noise_nodepth"]))
    assert np.array_equal(nd["z_vals"].numpy(), g["nodepth.z_vals"])
    close(nd["raw"], g["nodepth.raw"], rtol=2e-5, atol=2e-6)
    close(nd["depth"], g["nodepth.depth"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("name", ["scene_cfg1.npz", "scene_s75.npz", "scene_nd0.npz"])
@pytest.mark.parametrize("tag,emd", [("emd", 0.01), ("noemd", 0.0)])
def test_scene_train_and_grads_match_reference(name, tag, emd):
    g = load_golden(name)
    cfg = cfg_for(name)
    m = scene_from(g, cfg)
    ro = T(g["rays_o"]).clone().requires_grad_(True)
    rd = T(g["rays_d"]).clone().requires_grad_(True)
    ret = m.train_forward(ro, rd, T(g["target_rgb"]), T(g["target_d"]), T(g["noise"]), emd)
    for k in ("rgb", "depth", "rgb_loss", "depth_loss", "sdf_loss", "fs_loss"):
        close(ret[k], g[f"{tag}.{k}"], rtol=2e-5, atol=1e-7)
    close(ret["psnr"], g[f"{tag}.psnr"].reshape(()), rtol=1e-5)
    loss = path_cpu.total_loss(ret, cfg["training"])
    close(loss, g[f"{tag}.loss"], rtol=2e-5)
    loss.backward()
    close(ro.grad, g[f"{tag}.d_rays_o"], rtol=1e-3, atol=1e-5)
    close(rd.grad, g[f"{tag}.d_rays_d"], rtol=1e-3, atol=1e-5)
    for k, v in m.named_parameters():
        if v.numel():
            ref = g[f"{tag}.g.{k}"]
            scale = np.abs(ref).max() + 1e-12
            assert np.abs(v.grad.numpy() - ref).max() <= 2e-4 * scale, k


def test_decoder_matches_reference():
    g = load_golden("decoder.npz")
    w = {k[2:]: T(g[k]).clone().requires_grad_(True) for k in g.files if k.startswith("w.")}
    e = T(g["embed"]).clone().requires_grad_(True)
    pe = T(g["embed_pos"]).clone().requires_grad_(True)
    x = T(g["x"]).clone().requires_grad_(True)
    out = path_cpu.decoder_forward(w, e, pe, x)
    close(out, g["out"], rtol=1e-5, atol=1e-6)
    out.backward(T(g["gout"]))
    close(e.grad, g["d_embed"], rtol=1e-4, atol=1e-6)
    close(pe.grad, g["d_embed_pos"], rtol=1e-4, atol=1e-6)
    close(x.grad, g["d_x"], rtol=1e-4, atol=1e-6)
    for k, v in w.items():
        close(v.grad, g["g." + k], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag,emd", [("emd", 0.01), ("noemd", 0.0)])
def test_sdf_losses_match_reference(tag, emd):
    g = load_golden("losses.npz")
    s = T(g["sdf"]).clone().requires_grad_(True)
    p = T(g["prob"]).clone().requires_grad_(True)
    fs, sd = path_cpu.sdf_losses(T(g["z_vals"]), T(g["target_d"]), s, p, float(g["truncation"]), 5, emd)
    close(fs, g[f"{tag}_fs"], rtol=1e-6)
    close(sd, g[f"{tag}_sdf"], rtol=1e-6)
    (3.0 * fs + 7.0 * sd).backward()
    close(s.grad, g[f"{tag}_dsdf"], rtol=1e-5, atol=1e-9)
    if emd > 0:
        close(p.grad, g[f"{tag}_dprob"], rtol=1e-5, atol=1e-9)


def test_sdf_losses_no_depth_is_nan_like_reference():
    g = load_golden("losses.npz")
    fs, sd = path_cpu.sdf_losses(T(g["z_vals"]), torch.zeros(48, 1), T(g["sdf"]), T(g["prob"]), 0.1, 5, 0.01)
    assert math.isnan(float(g["nodepth_fs"])) == math.isnan(float(fs))
    assert math.isnan(float(g["nodepth_sdf"])) == math.isnan(float(sd))


def test_adam_restatement_matches_torch_optim():
    g = load_golden("adam.npz")
    for name, lr, eps, wd in (("grid", 0.01, 1e-15, 0.0), ("dec", 0.01, 1e-8, 1e-6)):
        p = T(g[f"{name}0"]).clone()
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        for s in range(g[f"{name}_grads"].shape[0]):
            p, m, v = path_cpu.adam_reference(p, T(g[f"{name}_grads"][s]), m, v, s + 1, lr, 0.9, 0.99, eps, wd)
            close(p, g[f"{name}_traj"][s], rtol=2e-5, atol=1e-8)
        close(m, g[f"{name}_m"], rtol=1e-5, atol=1e-10)
        close(v, g[f"{name}_v"], rtol=1e-5, atol=1e-13)


def test_quaternion_helpers_match_golden():
    g = load_golden("quaternion.npz")
    rot = T(g["rot"]).clone().requires_grad_(True)
    trans = T(g["trans"]).clone().requires_grad_(True)
    R = p3d_cpu.quaternion_to_matrix(rot)
    Tm = torch.eye(4)[None].repeat(rot.shape[0], 1, 1)
    Tm[:, :3, :3] = R
    Tm[:, :3, 3] = trans
    close(Tm, g["T"], rtol=1e-6, atol=1e-7)
    Tm.backward(T(g["gT"]))
    close(rot.grad, g["d_rot"], rtol=1e-5, atol=1e-6)
    close(trans.grad, g["d_trans"], rtol=1e-6)
    close(p3d_cpu.matrix_to_quaternion(T(g["T"])[:, :3, :3]), g["q_back"], rtol=1e-6, atol=1e-7)


def test_hashgrid_oracle_self_consistency():
    """Explicit tcnn-style backward == autograd of a differentiable re-derivation."""
    g = load_golden("hashgrid.npz")
    meta = tcnn_cpu.make_grid_meta(16, 2, 10, 16, float(2.0 ** (math.log2(16) / 15)))
    assert np.array_equal(np.array(meta.offsets), g["t10.offsets"])
    assert np.array_equal(np.array(meta.scales, dtype=np.float32), g["t10.scales"])
    x = T(g["t10.x"])[5:200].clone()
    params = T(g["t10.params"]).clone()
    dy = T(g["t10.dy"])[5:200]
    # differentiable version: weights from frac as torch ops, gather with fixed integer indices
    xd = x.clone().requires_grad_(True)
    pd = params.clone().requires_grad_(True)
    table = pd.view(-1, 2)
    outs = []
    for level in range(16):
        scale = meta.scales[level]
        pos = xd * scale + 0.5
        fl = torch.floor(pos).detach()
        frac = pos - fl
        cell = fl.to(torch.int64)
        acc = 0
        for c in range(8):
            w = 1.0
            corner = []
            for d in range(3):
                bit = (c >> d) & 1
                w = w * (frac[:, d] if bit else 1 - frac[:, d])
                corner.append(cell[:, d] + bit)
            idx = tcnn_cpu._grid_index(torch.stack(corner, -1) & 0xFFFFFFFF,
                                       meta.offsets[level + 1] - meta.offsets[level], meta.resolutions[level])
            acc = acc + w[:, None] * table[idx + meta.offsets[level]]
        outs.append(acc)
    y = torch.cat(outs, -1)
    close(y, tcnn_cpu.hashgrid_forward(x, params, meta), rtol=1e-4, atol=3e-5)  # non-fused pos differs by 1 ulp
    y.backward(dy)
    dparams, dx = tcnn_cpu.hashgrid_backward(x, params, dy, meta)
    close(dparams, pd.grad, rtol=1e-3, atol=1e-4)
    close(dx, xd.grad, rtol=1e-3, atol=1e-3)


def test_hashgrid_level_table_matches_survey():
    meta = tcnn_cpu.make_grid_meta(16, 2, 19, 16, float(2.0 ** (math.log2(16) / 15)))
    sizes = np.diff(np.array(meta.offsets))
    assert list(sizes[:9]) == [4096, 8000, 13824, 21952, 39304, 68928, 117656, 205384, 357912]
    assert all(s == 524288 for s in sizes[9:])
    assert meta.n_params == 9014144
    meta16 = tcnn_cpu.make_grid_meta(16, 2, 16, 16, float(2.0 ** (math.log2(16) / 15)))
    assert meta16.offsets[-1] == 808072


def test_hashgrid_indices_in_range_and_x_pairs():
    meta = tcnn_cpu.make_grid_meta(16, 2, 19, 16, float(2.0 ** (math.log2(16) / 15)))
    x = torch.rand(500, 3)
    idx = tcnn_cpu.hashgrid_indices(x, meta)
    sizes = torch.tensor(np.diff(np.array(meta.offsets)))
    assert (idx >= 0).all() and (idx < sizes[None, :, None]).all()


def test_frequency_oracle_values():
    x = torch.tensor([[0.0, 0.25, 1.0]])
    y = tcnn_cpu.frequency_forward(x, 2)
    # dim 1 (x=0.25): sin(pi/4), cos(pi/4), sin(pi/2), cos(pi/2)
    close(y[0, 4:8], [math.sin(math.pi / 4), math.cos(math.pi / 4), 1.0, 0.0], atol=1e-6)
    assert y.shape == (1, 12)


@pytest.mark.skipif(not ref_import.available(), reason="reference tree only exists in the build container")
def test_oracle_equals_live_reference_on_fresh_seed():
    """Beyond the committed vectors: a fresh seed, live against the reference's Python."""
    ref = ref_import.load()
    cfg = synth.config_plumbing()
    torch.manual_seed(123)
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    model = ref.scene_rep.JointEncoding(cfg, bb, nf)
    with torch.no_grad():
        model.embed_fn.params.copy_(torch.randn_like(model.embed_fn.params) * 0.2)
    mine = path_cpu.CpuScene(cfg, bb, nf)
    mine.load_state_dict(model.state_dict())
    N, S = 64, 16
    ro = torch.rand(N, 3) * 0.2
    rd = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    rgb, d = torch.rand(N, 3), torch.rand(N, 1) * 1.5
    d[:3] = 0
    noise = torch.rand(N, S)
    orig = torch.rand
    torch.rand = lambda *a, **k: noise.clone()
    try:
        model.train()
        r = model.forward(ro, rd, rgb, d)
    finally:
        torch.rand = orig
    o = mine.train_forward(ro, rd, rgb, d, noise, 0.01)
    for k in ("rgb", "depth", "rgb_loss", "depth_loss", "sdf_loss", "fs_loss"):
        close(o[k].detach(), r[k].detach(), rtol=2e-5, atol=1e-7)


def test_plain_c_restatement_agrees_bit_for_bit():
    """oracle/c/hashgrid_ref.c (real uint32 wrap, real fmaf) == oracle/tcnn_cpu.py (int64-masked, fp64-emulated)."""
    import subprocess
    from oracle import c_ref
    from .conftest import ROOT
    import os
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "c")], check=True, capture_output=True)
    meta = tcnn_cpu.make_grid_meta(16, 2, 19, 16, float(2.0 ** (math.log2(16) / 15)))
    torch.manual_seed(0)
    x = torch.rand(3000, 3)
    x[0], x[1], x[2] = 0.0, 1.0, torch.tensor([-0.05, 1.07, 0.5])
    params = torch.rand(meta.n_params) * 2 - 1
    idx, y = c_ref.hashgrid(x.numpy(), params.numpy(), meta)
    assert np.array_equal(idx.astype(np.int64), tcnn_cpu.hashgrid_indices(x, meta).numpy())
    assert np.array_equal(y, tcnn_cpu.hashgrid_forward(x, params, meta).numpy())
    g = load_golden("hashgrid.npz")
    idx19, _ = c_ref.hashgrid(g["t19.x"], None, meta)
    assert np.array_equal(idx19.astype(np.int32), g["t19.idx"])


# ------------------------------------------------------------------------------ RandomOptimizer
def test_random_optimizer_restatement_matches_reference():
    """oracle/ro_cpu.py vs poses produced by the reference's own RandomOptimizer.optimize (tests/golden/ro.npz)."""
    from oracle import ro_cpu
    g = load_golden("ro.npz")
    m = scene_from(g, synth.config_plumbing())
    m.eval()
    pst, rows, cols = T(g["pst"]), T(g["rows"]).long(), T(g["cols"]).long()
    depth, rays_dir, init = T(g["depth"]), T(g["rays_dir"]), T(g["init_pose"])
    c1, c2, trunc = float(g["c1"]), float(g["c2"]), float(g["trunc"])
    # one fitness evaluation
    pst7 = ro_cpu.pose_6d_to_7d(pst * c1)
    # one ulp of fp32 at |v| <= 1: torch's CPU sin / cos / sqrt kernels differ by that much between the SIMD paths of
    # different hosts (bit-equal on the machine that wrote the fixture, 26 of 672 values 6e-8 off on the GPU hosts' CPUs)
    close(pst7, g["pst7_0"], rtol=0, atol=1.2e-7)
    td = depth[rows, cols][:, None]
    world, a_rot, a_trans = ro_cpu.particle_points(init[:3, :3], init[:3, 3:], pst7, rays_dir[rows, cols, :] * td)
    close(a_rot, g["abs_rot0"], rtol=1e-6, atol=1e-7)
    close(a_trans, g["abs_trans0"], rtol=1e-6, atol=1e-7)
    with torch.no_grad():
        mms = ro_cpu.mean_masked_sdf(m.run_network, world, td, trunc)
    close(mms, g["mean_masked0"], rtol=2e-5, atol=1e-7)
    # whole optimisation, pose after every round
    for n_iter in range(0, 7):
        pose, trace = ro_cpu.optimize(m.run_network, pst, rows, cols, depth, rays_dir, init, n_iter, c1, c2, trunc)
        close(pose, g[f"pose_after_{n_iter}"], rtol=1e-4, atol=2e-6)
        assert len(trace) == n_iter


# --------------------------------------------------------------------- BASELINE config 3 in miniature (CPU)
def oracle_backend():
    """tests/seq_harness.py Backend over the ORACLE (path_cpu.CpuScene, ro_cpu, p3d_cpu) and the product's
    host-side samplers / keyframe-ray index logic (CPU tensors only; nothing is launched)."""
    import copy
    import types
    from oracle import ro_cpu
    from mipsfusion_amd.helper_functions import sampling_helper as sh

    class OracleModel(path_cpu.CpuScene):
        def __init__(self, cfg, bb, nf):
            super().__init__(cfg, bb, nf)
            self.initial_dict = copy.deepcopy(self.state_dict())

        def recover_initial_param(self):
            self.load_state_dict(self.initial_dict)

        def forward(self, rays_o, rays_d, target_rgb, target_d, EMD_w=0.01):
            S = self.cfg["training"]["n_samples_d"] + self.cfg["training"]["n_range_d"]
            return self.train_forward(rays_o, rays_d, target_rgb, target_d, torch.rand(rays_o.shape[0], S), EMD_w)

    class KfSet:      # model/keyframeSet.py:25, 76-79, 170-175, 386-455 on CPU tensors
        def __init__(self, cfg, H, W, num_kf):
            s = cfg["sampling"]
            self.rows, self.cols = sh.sample_pixels_uniformly(H, W, s["kf_n_rays_h"], s["kf_n_rays_w"])
            self.R = s["kf_n_rays_h"] * s["kf_n_rays_w"]
            self.rays = torch.zeros(num_kf, self.R, 7)
            self.n = 0

        def add_keyframe(self, frame):
            rays = torch.cat([frame["direction"], frame["rgb"], frame["depth"][..., None]], -1)
            self.rays[self.n] = rays[self.rows, self.cols]
            self.n += 1

        def _db(self):
            from mipsfusion_amd.keyframe_rays import DeviceRayDB
            db = DeviceRayDB.__new__(DeviceRayDB)
            db.num_rays_to_save, db.device, db.rays = self.R, torch.device("cpu"), self.rays
            db._gather = lambda flat: self.rays.reshape(-1, 7)[flat]       # the gather kernel's job, on the host
            return db

        def sample_rays_in_submap(self, *a):
            return self._db().sample_rays_in_submap(*a)

        def sample_rays_in_given_kf(self, *a):
            return self._db().sample_rays_in_given_kf(*a)

    def make_ro(cfg, slam):
        r = cfg["tracking"]["RO"]
        pst = torch.from_numpy(np.random.multivariate_normal(np.zeros(6), np.eye(6), r["particle_size"]).astype(np.float32))
        pst[0, :] = 0
        rows, cols = sh.sample_pixels_uniformly(slam.dataset.H, slam.dataset.W, r["n_rows"], r["n_cols"])
        return types.SimpleNamespace(pst=torch.clamp(pst, -2., 2.), rows=rows, cols=cols, dirs=slam.dataset.rays_d,
                                     c1=r["initial_scaling_factor"], c2=r["rescaling_factor"], trunc=cfg["training"]["trunc"])

    def ro_optimize(ro, model, depth, init, last, n):
        return ro_cpu.optimize(model.run_network, ro.pst, ro.rows, ro.cols, depth, ro.dirs, init, n, ro.c1, ro.c2,
                               ro.trunc)[0]

    return types.SimpleNamespace(
        device=torch.device("cpu"), make_model=OracleModel, deepcopy=copy.deepcopy, Adam=torch.optim.Adam, sh=sh,
        qt_to_transform_matrix=p3d_cpu.qt_to_transform_matrix if hasattr(p3d_cpu, "qt_to_transform_matrix") else None,
        matrix_to_quaternion=p3d_cpu.matrix_to_quaternion, make_kfset=KfSet, make_ro=make_ro, ro_optimize=ro_optimize)


def test_two_submap_sequence_oracle_matches_reference_run():
    """The oracle + the product's HOST logic (pixel samplers, keyframe-ray index generation) driven through the
    config-3 loop of tests/seq_harness.py reproduce the run over the reference's own classes (sequence.npz):
    index stream bit for bit, losses and poses to fp32 round-off."""
    from mipsfusion_amd.helper_functions import geometry_helper as gh
    from . import seq_harness
    g = load_golden("sequence.npz")
    B = oracle_backend()
    B.qt_to_transform_matrix = gh.qt_to_transform_matrix
    out = seq_harness.run_sequence(B)
    assert list(out["tags"]) == [str(t) for t in g["tags"]]
    assert np.array_equal(torch.cat(out["idx"]).numpy(), g["idx_flat"])
    np.testing.assert_allclose(out["losses"], g["losses"], rtol=2e-3)
    np.testing.assert_allclose(out["est"], g["est"], atol=2e-4)
    for sm in (0, 1):
        close(out["models"][sm]["decoder.sdf_linear.2.weight"], g[f"m{sm}.decoder.sdf_linear.2.weight"], rtol=2e-2,
              atol=2e-3)
