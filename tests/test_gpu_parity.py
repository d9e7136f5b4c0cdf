"""GPU (-m gpu): the HIP path, called through the C ABI, against (a) the committed golden vectors the reference
produced, (b) the CPU oracle on fresh seeded inputs, (c) size-independent properties at the full BASELINE sizes.

Tolerances: integer/index results bit-exact; sample placement (z_vals) bit-exact; floating point within 1e-4
relative (BASELINE.json north_star) -- most checks are tighter and say so."""
import math
import os
import random

import numpy as np
import pytest
import torch

from mipsfusion_amd import _lib, ops, synth
from mipsfusion_amd.model import JointEncoding, MLP_reg, get_encoder
from mipsfusion_amd.optim import FusedAdam
from oracle import path_cpu, tcnn_cpu

from .conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
PLS = float(2.0 ** (math.log2(256 / 16) / 15))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the gpu-marked tests must run on the MI355X box")
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_err(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: max error relative to max magnitude {e:.3e} > {tol:.1e}"


def assert_grad_close(a, b, tol, what="", outlier_frac=1e-4):
    """Gradients pass through ReLU / first-crossing / band masks: two fp32 implementations with different summation
    orders legitimately flip O(1) of millions of such decisions, which moves a handful of entries.  Require the
    relative L2 error and all but `outlier_frac` of the entries to be within tolerance."""
    a = a.detach().double().cpu().numpy().ravel()
    b = b.detach().double().cpu().numpy().ravel() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64).ravel()
    scale = np.abs(b).max() + 1e-30
    bad = np.abs(a - b) > tol * scale
    l2 = np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)
    assert bad.mean() <= outlier_frac and l2 <= 4 * tol, \
        f"{what}: {bad.sum()} of {bad.size} entries off by > {tol:.0e}*max, relative L2 error {l2:.3e}"


def test_device_is_gfx950(dev):
    assert _lib.lib().mipsf_device_cu_count() >= 64
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ------------------------------------------------------------------------------ hash grid
@pytest.mark.parametrize("tag,log2_t", [("t10", 10), ("t19", 19)])
def test_hashgrid_golden(dev, tag, log2_t):
    g = load_golden("hashgrid.npz")
    meta = _lib.make_grid_meta(16, 2, log2_t, 16, PLS)
    gen = torch.Generator().manual_seed(int(g[f"{tag}.param_seed"]))
    params = ((torch.rand(meta.n_params, generator=gen) * 2 - 1) * 0.5).to(dev)
    x = T(g[f"{tag}.x"]).to(dev)
    idx = ops.hashgrid_indices(x, meta).cpu().numpy()
    assert np.array_equal(idx, g[f"{tag}.idx"]), "hash-grid corner indices must be bit-exact"
    y = ops.hashgrid_fwd(x, params, meta, _lib.FEAT_AOS)
    assert_close(y, g[f"{tag}.y"], 2e-6, "features (AoS)")
    ylm = ops.hashgrid_fwd(x, params, meta, _lib.FEAT_LEVEL_MAJOR)
    assert torch.equal(ylm.permute(1, 0, 2).reshape(x.shape[0], 32), y), "layouts must hold identical values"
    dy = T(g[f"{tag}.dy"]).to(dev)
    dparams = torch.zeros_like(params)
    dx = torch.zeros_like(x)
    ops.hashgrid_bwd(x, params, dy, dparams, meta, _lib.FEAT_AOS, dx)
    nz = T(g[f"{tag}.dparams_nz_idx"]).long()
    assert_close(dparams.cpu()[nz], g[f"{tag}.dparams_nz_val"], 1e-5, "dL/dparams (touched entries)")
    mask = torch.ones(meta.n_params, dtype=torch.bool)
    mask[nz] = False
    assert float(dparams.cpu()[mask].abs().max()) == 0.0, "untouched entries must stay exactly zero"
    assert_close(dx, g[f"{tag}.dx"], 1e-4, "dL/dx")
    # level-major backward gives the same gradients
    dparams2 = torch.zeros_like(params)
    dx2 = torch.zeros_like(x)
    ops.hashgrid_bwd(x, params, dy.reshape(-1, 16, 2).permute(1, 0, 2).contiguous(), dparams2, meta,
                     _lib.FEAT_LEVEL_MAJOR, dx2)
    assert_close(dparams2, dparams, 1e-5, "level-major dparams")
    assert_close(dx2, dx, 1e-5, "level-major dx")
    # Jacobian-saving forward: same features, and dL/dx from the saved Jacobian is bit-identical to the gather path
    for layout, dyl, ref in ((_lib.FEAT_AOS, dy, dx), (_lib.FEAT_LEVEL_MAJOR,
                                                       dy.reshape(-1, 16, 2).permute(1, 0, 2).contiguous(), dx2)):
        yj, jac = ops.hashgrid_fwd(x, params, meta, layout, with_jac=True)
        assert torch.equal(yj, y if layout == _lib.FEAT_AOS else ylm), "fwd_jac features"
        dxj = torch.zeros_like(x)
        ops.hashgrid_dx_from_jac(jac, dyl, dxj, meta, layout)
        assert torch.equal(dxj, ref), "dx from the saved Jacobian must equal the gather path bit for bit"


def test_hashgrid_vs_oracle_fresh(dev):
    torch.manual_seed(21)
    meta = _lib.make_grid_meta(16, 2, 19, 16, PLS)
    ometa = tcnn_cpu.make_grid_meta(16, 2, 19, 16, PLS)
    M = 6000                                      # ragged: not a multiple of 64/256
    x = torch.rand(M, 3)
    x[:50] = torch.rand(50, 1).expand(50, 3) * torch.tensor([1.0, 0.0, 0.0]) + torch.tensor([0.0, 0.3, 0.7])  # one ray
    params = (torch.rand(ometa.n_params) * 2 - 1) * 0.1
    dy = torch.randn(M, 32)
    y_ref = tcnn_cpu.hashgrid_forward(x, params, ometa)
    idx_ref = tcnn_cpu.hashgrid_indices(x, ometa)
    dp_ref, dx_ref = tcnn_cpu.hashgrid_backward(x, params, dy, ometa)
    xg, pg = x.to(dev), params.to(dev)
    assert np.array_equal(ops.hashgrid_indices(xg, meta).cpu().numpy(), idx_ref.numpy().astype(np.int32))
    assert_close(ops.hashgrid_fwd(xg, pg, meta), y_ref, 2e-6, "features")
    dp = torch.zeros_like(pg)
    dx = torch.zeros_like(xg)
    ops.hashgrid_bwd(xg, pg, dy.to(dev), dp, meta, _lib.FEAT_AOS, dx)
    assert_close(dp, dp_ref, 2e-5, "dparams")
    assert_close(dx, dx_ref, 1e-4, "dx")



@pytest.mark.parametrize("seed", list(range(10)))
def test_hashgrid_backward_fuzz_exact_zero_pattern(dev, seed):
    """Randomised shapes and point sets for the routed scatter, checked the way that found the routing fault of round 6: close
    to the oracle AND an entry the oracle never touches is exactly zero.  Table 2^10 .. 2^21, 1 .. 40 000 points mixing the box's
    interior, both outsides, exact 0 / 1 coordinates, exact cell boundaries of a random level, one crowded cell, duplicated
    points; gradients spread over 2^40 with dead ray tails, dead samples and dead (sample, level) pairs; both layouts."""
    g = torch.Generator().manual_seed(1000 + seed)
    log2_t = [10, 14, 16, 19, 21, 12, 19, 16, 19, 17][seed]
    M = int(torch.randint(1, 40001, (1,), generator=g)) if seed else 1
    meta = _lib.make_grid_meta(16, 2, log2_t, 16, PLS)
    ometa = tcnn_cpu.make_grid_meta(16, 2, log2_t, 16, PLS)
    x = torch.rand(M, 3, generator=g)
    kind = torch.randint(0, 8, (M,), generator=g)
    x[kind == 1] = x[kind == 1] * 1.4 - 0.2                                          # around the box, both sides
    x[kind == 2] = -torch.rand(int((kind == 2).sum()), 3, generator=g) * 1e-3       # just below zero
    x[kind == 3] = torch.randint(0, 2, (int((kind == 3).sum()), 3), generator=g).float()      # corners of the box: exact 0 / 1
    lvl = int(torch.randint(0, 16, (1,), generator=g))
    sc = float(ometa.scales[lvl])                  # pos = scale * x + 0.5: x = (k - 0.5) / scale sits ON a cell boundary of that level
    n4 = int((kind == 4).sum())
    x[kind == 4] = (torch.randint(0, int(ometa.resolutions[lvl]), (n4, 3), generator=g).float() - 0.5) / sc
    if M > 500:
        x[100:400] = torch.tensor([0.77, 0.13, 0.52]) + torch.rand(300, 3, generator=g) * 1e-5           # one crowded cell
        x[400:450] = x[400]                                                                                # duplicates
    dy = torch.randn(M, 32, generator=g) * torch.exp2(-40.0 * torch.rand(M, 1, generator=g))
    tail = torch.randint(0, 65, ((M + 63) // 64,), generator=g).repeat_interleave(64)[:M]
    dy[(torch.arange(M) % 64) >= tail] = 0.0
    dy[torch.rand(M, generator=g) < 0.1] = 0.0
    dy[(torch.rand(M, 16, generator=g) < 0.1)[:, :, None].expand(M, 16, 2).reshape(M, 32)] = 0.0
    params = torch.zeros(ometa.n_params)
    dp_ref, _ = tcnn_cpu.hashgrid_backward(x, params, dy, ometa, need_dx=False)
    touched = dp_ref != 0
    for lay, d in ((_lib.FEAT_AOS, dy), (_lib.FEAT_LEVEL_MAJOR, dy.reshape(M, 16, 2).permute(1, 0, 2).contiguous())):
        dp = torch.zeros(ometa.n_params, device=dev)
        ops.hashgrid_bwd(x.to(dev), params.to(dev), d.to(dev), dp, meta, lay, None)
        got = dp.cpu()
        if float(dp_ref.abs().max()) > 0:
            assert_close(got, dp_ref, 2e-5, f"seed {seed}: dparams T=2^{log2_t} M={M}")
        assert float(got[~touched].abs().max()) == 0.0, f"seed {seed}: a gradient in an entry no sample touches (T=2^{log2_t}, M={M})"


@pytest.mark.parametrize("log2_t", [19, 16])
def test_hashgrid_backward_points_outside_the_box(dev, log2_t):
    """Samples outside the bounding box (normalised coordinates below 0 or above 1: rays run past the box, scene_rep.py:134-146
    normalises without clamping, tcnn floors and wraps).  A point just below 0 has cell coordinate (uint32_t)-1 on every level;
    until round 6 the routing kernel's fast path for power-of-two hashed levels took `cx + 1 < 2^13` (0 after the wrap) as
    proof that the cell's x-pairs stay inside one table slice, and the second corner of such a pair was added OUTSIDE the
    slice's LDS image -- lost, or (planar slice layout) visible as a tiny gradient of an entry nobody touched.  Gradient
    magnitudes spread over 2^40 like a mapping step's: a stray 1e-11 beside 1e-6 entries passes any relative gate, so the
    check is the exact one -- an entry the oracle leaves untouched stays exactly zero, a touched one is touched."""
    torch.manual_seed(7 + log2_t)
    M = 60000
    meta = _lib.make_grid_meta(16, 2, log2_t, 16, PLS)
    ometa = tcnn_cpu.make_grid_meta(16, 2, log2_t, 16, PLS)
    x = torch.rand(M, 3) * 1.2 - 0.1                                  # a sixth of the points outside, on either side
    x[:2000, 0] = -torch.rand(2000) * 1e-3                            # just below zero in x: the wrapping cell coordinate
    x[2000:3000, 1] = -torch.rand(1000) * 1e-3
    x[3000:4000, 2] = 1.0 + torch.rand(1000) * 1e-3
    dy = torch.randn(M, 32) * torch.exp2(-40.0 * torch.rand(M, 1))
    params = torch.zeros(ometa.n_params)
    dp_ref, _ = tcnn_cpu.hashgrid_backward(x, params, dy, ometa, need_dx=False)
    for lay, d in ((_lib.FEAT_AOS, dy), (_lib.FEAT_LEVEL_MAJOR, dy.reshape(M, 16, 2).permute(1, 0, 2).contiguous())):
        dp = torch.zeros(ometa.n_params, device=dev)
        ops.hashgrid_bwd(x.to(dev), params.to(dev), d.to(dev), dp, meta, lay, None)
        got = dp.cpu()
        assert_close(got, dp_ref, 2e-5, f"dparams with points outside the box, T=2^{log2_t}")
        touched = dp_ref != 0
        assert float(got[~touched].abs().max()) == 0.0, "a gradient in an entry no sample touches"
        # (an entry whose contributions cancel exactly in the oracle's fp32 order may keep a residual here and vice versa: rare)
        assert int(((got == 0) & touched).sum()) <= 4, "a touched entry without a gradient"
    # the same points through the gather side: corner indices bit-exact, features, d x (gather path and saved Jacobian)
    n = 6000
    xs, pr = x[:n].contiguous(), (torch.rand(ometa.n_params) * 2 - 1) * 0.1
    dys = torch.randn(n, 32)
    assert np.array_equal(ops.hashgrid_indices(xs.to(dev), meta).cpu().numpy(), tcnn_cpu.hashgrid_indices(xs, ometa).numpy().astype(np.int32))
    assert_close(ops.hashgrid_fwd(xs.to(dev), pr.to(dev), meta), tcnn_cpu.hashgrid_forward(xs, pr, ometa), 2e-6, "features outside the box")
    _, dx_ref = tcnn_cpu.hashgrid_backward(xs, pr, dys, ometa)
    dxg, dpg = torch.zeros(n, 3, device=dev), torch.zeros(ometa.n_params, device=dev)
    ops.hashgrid_bwd(xs.to(dev), pr.to(dev), dys.to(dev), dpg, meta, _lib.FEAT_AOS, dxg)
    assert_close(dxg, dx_ref, 1e-4, "d x outside the box")
    _, jac = ops.hashgrid_fwd(xs.to(dev), pr.to(dev), meta, _lib.FEAT_AOS, with_jac=True)
    dxj = torch.zeros(n, 3, device=dev)
    ops.hashgrid_dx_from_jac(jac, dys.to(dev), dxj, meta, _lib.FEAT_AOS)
    assert torch.equal(dxj, dxg), "d x from the saved Jacobian = the gather path, outside the box too"

@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize("log2_t,M", [(16, 1), (16, 70000), (22, 3001), (19, 40000)])
def test_hashgrid_backward_routing_edge_sizes(dev, log2_t, M, sparse):
    """The routed scatter at its edges: a single sample; a batch larger than one 32768-record part on a 2^16 table
    (every bin multi-part -> partial slices + reduce); a 2^22 table (410 slices per level, 3000 bins); the headline 2^19
    table -- all against the oracle, with a degenerate cluster (many samples in one cell: run merging + same-address
    atomics).  sparse: the feature gradient a mapping step produces -- exactly zero on the tail of every 64-sample ray
    (a different tail length per ray), on scattered whole samples and on scattered single (sample, level) pairs: the
    routing kernel makes no record for a zero pair and packs the live samples of a workgroup before ranking them
    (scatter_route_kernel); both layouts."""
    torch.manual_seed(30 + log2_t)
    meta = _lib.make_grid_meta(16, 2, log2_t, 16, PLS)
    ometa = tcnn_cpu.make_grid_meta(16, 2, log2_t, 16, PLS)
    x = torch.rand(M, 3)
    if M > 2000:
        x[100:1100] = torch.tensor([0.4312, 0.2521, 0.8133]) + torch.rand(1000, 3) * 1e-4     # one fine cell
    dy = torch.randn(M, 32)
    if sparse:
        s_in_ray = torch.arange(M) % 64
        tail = torch.randint(0, 65, ((M + 63) // 64,)).repeat_interleave(64)[:M]               # live samples per ray: 0..64
        dy[s_in_ray >= tail] = 0.0
        dy[torch.rand(M) < 0.1] = 0.0                                                          # whole samples
        pair_dead = (torch.rand(M, 16) < 0.1)[:, :, None].expand(M, 16, 2).reshape(M, 32)      # single (sample, level) pairs
        dy[pair_dead] = 0.0
        if M == 1:
            dy[0, 4:6] = 0.0
    params = torch.zeros(ometa.n_params)
    dp_ref, _ = tcnn_cpu.hashgrid_backward(x, params, dy, ometa, need_dx=False)
    dp = torch.zeros(ometa.n_params, device=dev)
    ops.hashgrid_bwd(x.to(dev), params.to(dev), dy.to(dev), dp, meta, _lib.FEAT_AOS, None)
    assert_close(dp, dp_ref, 2e-5, f"dparams T=2^{log2_t} M={M}")
    touched = dp_ref != 0
    assert float(dp.cpu()[~touched].abs().max()) == 0.0
    # accumulate semantics: a second call adds on top
    ops.hashgrid_bwd(x.to(dev), params.to(dev), dy.to(dev), dp, meta, _lib.FEAT_AOS, None)
    assert_close(dp, 2 * dp_ref, 2e-5, "accumulation into dparams")
    # level-major layout of the same gradient (the layout the fused scene path uses)
    dp_lm = torch.zeros(ometa.n_params, device=dev)
    ops.hashgrid_bwd(x.to(dev), params.to(dev), dy.reshape(M, 16, 2).permute(1, 0, 2).contiguous().to(dev), dp_lm, meta,
                     _lib.FEAT_LEVEL_MAJOR, None)
    assert_close(dp_lm, dp_ref, 2e-5, "level-major dparams")
    # MIPSF_HG_DPARAMS_ZERO: a buffer the caller vouches to be zero takes stores instead of read-modify-writes -- same bits
    dp_z = torch.zeros(ometa.n_params, device=dev)
    ops.hashgrid_bwd(x.to(dev), params.to(dev), dy.reshape(M, 16, 2).permute(1, 0, 2).contiguous().to(dev), dp_z, meta,
                     _lib.FEAT_LEVEL_MAJOR, None, dparams_zero=True)
    assert_close(dp_z, dp_ref, 2e-5, "dparams into a buffer known to be zero")
    assert float((dp_z - dp_lm).abs().max()) <= 1e-6 * float(dp_ref.abs().max()) + 1e-30
    if sparse and M > 1:        # an all-zero gradient: no records at all, nothing written
        dp0 = torch.zeros(ometa.n_params, device=dev)
        ops.hashgrid_bwd(x.to(dev), params.to(dev), torch.zeros(M, 32, device=dev), dp0, meta, _lib.FEAT_AOS, None)
        assert float(dp0.abs().max()) == 0.0


def test_scatter_counter_block_survives_many_alternating_calls(dev):
    """The routed scatter keeps a small counter block between calls (bin counts, queue head, two tickets) that every call
    must leave ready: 600 calls alternating between two batches of different size and sparsity while a second stream keeps
    the memory system busy; every result must equal the first call's for the same batch to the scatter's own
    reproducibility (fp64 slice sums: differences below 1e-6 of the maximum), and the block must read zero (counts, head,
    tickets) after every call."""
    torch.manual_seed(41)
    meta = _lib.make_grid_meta(16, 2, 19, 16, PLS)
    params = torch.zeros(meta.n_params, device=dev)
    batches = []
    for M in (20000, 4096 * 8 + 13):
        x = torch.rand(M, 3, device=dev)
        dy = torch.randn(16, M, 2, device=dev)
        dy[:, torch.rand(M, device=dev) < 0.5] = 0.0
        batches.append((x, dy))
    counters = ops._scatter_counters(dev, meta)
    n_bins_plus = _lib.buffer_size(_lib.SIZE_HASHGRID_COUNTER_WORDS, meta=meta) - 8
    n_bins = n_bins_plus // 3
    want = []
    for x, dy in batches:
        dp = torch.zeros(meta.n_params, device=dev)
        ops.hashgrid_bwd(x, params, dy, dp, meta, _lib.FEAT_LEVEL_MAJOR, None)
        want.append(dp)
    busy = torch.empty(64 << 20, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for q in range(200):
            busy[: (1 + q % 7) << 22].add_(1.0)
    dp = torch.zeros(meta.n_params, device=dev)
    worst = 0.0
    for it in range(600):
        k = it & 1
        dp.zero_()
        ops.hashgrid_bwd(batches[k][0], params, batches[k][1], dp, meta, _lib.FEAT_LEVEL_MAJOR, None)
        if it % 50 == 0 or it > 590:
            assert int(counters[: n_bins + 4].abs().sum()) == 0, f"counter block not left ready after call {it}"
            worst = max(worst, rel_err(dp, want[k]))
    torch.cuda.synchronize()
    assert worst < 1e-6, f"a call saw another call's counters: {worst:.2e}"


def test_hashgrid_empty_and_single(dev):
    meta = _lib.make_grid_meta(16, 2, 10, 16, PLS)
    params = torch.rand(meta.n_params, device=dev)
    assert ops.hashgrid_fwd(torch.empty(0, 3, device=dev), params, meta).shape == (0, 32)
    y = ops.hashgrid_fwd(torch.full((1, 3), 0.5, device=dev), params, meta)
    assert y.shape == (1, 32) and torch.isfinite(y).all()


def test_encoding_modules_autograd(dev):
    g = load_golden("hashgrid.npz")
    enc, dim = get_encoder("HashGrid", log2_hashmap_size=10, desired_resolution=256)
    enc = enc.to(dev)
    with torch.no_grad():
        enc.params.copy_(T(g["t10.params"]))
    x = T(g["t10.x"]).to(dev).requires_grad_(True)
    y = enc(x.double())                               # input cast to fp32 inside, like tcnn's binding
    y.backward(T(g["t10.dy"]).to(dev))
    assert_close(y, g["t10.y"], 2e-6, "module forward")
    assert_close(x.grad, g["t10.dx"], 1e-4, "module dx")
    freq, fdim = get_encoder("Frequency", n_bins=8)
    xf = T(g["freq.x"]).to(dev).requires_grad_(True)
    yf = freq.to(dev)(xf)
    assert_close(yf, g["freq.y8"], 2e-6, "frequency forward")
    yf.backward(T(g["freq.dy"]).to(dev))
    assert_close(xf.grad, g["freq.dx8"], 2e-5, "frequency dx")


# -------------------------------------------------------------------------------- decoder
def load_decoder(g, dev, prefix="w."):
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48)
    dec.load_state_dict({k[len(prefix):]: T(g[k]) for k in g.files if k.startswith(prefix)})
    return dec.to(dev)


def test_decoder_module_golden(dev):
    g = load_golden("decoder.npz")
    dec = load_decoder(g, dev)
    e = T(g["embed"]).to(dev).requires_grad_(True)
    pe = T(g["embed_pos"]).to(dev).requires_grad_(True)
    x = T(g["x"]).to(dev).requires_grad_(True)
    out = dec(e, pe, x)
    assert_close(out, g["out"], 1e-5, "decoder out")
    for c, name in enumerate(["r", "g", "b", "sdf", "entropy", "p0", "p1", "p2", "p3", "p4"]):
        assert_close(out[:, c], g["out"][:, c], 2e-5, f"decoder out column {name}")
    out.backward(T(g["gout"]).to(dev))
    assert_close(e.grad, g["d_embed"], 1e-4, "d embed")
    assert_close(pe.grad, g["d_embed_pos"], 1e-4, "d embed_pos")
    assert_close(x.grad, g["d_x"], 1e-4, "d x")
    for k, v in dec.named_parameters():
        assert_close(v.grad, g["g." + k], 1e-4, "grad " + k)


@pytest.mark.parametrize("M", [1, 31, 33, 128, 129, 1000])
def test_decoder_ragged_sizes_vs_oracle(dev, M):
    g = load_golden("decoder.npz")
    dec = load_decoder(g, dev)
    w = {k[2:]: T(g[k]) for k in g.files if k.startswith("w.")}
    torch.manual_seed(M)
    e, pe, x = torch.randn(M, 32) * 0.2, torch.rand(M, 48) * 2 - 1, torch.rand(M, 3)
    gout = torch.randn(M, 10)
    wr = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    er, per, xr = (t.clone().requires_grad_(True) for t in (e, pe, x))
    ref = path_cpu.decoder_forward(wr, er, per, xr)
    ref.backward(gout)
    eg, peg, xg = (t.to(dev).requires_grad_(True) for t in (e, pe, x))
    out = dec(eg, peg, xg)
    out.backward(gout.to(dev))
    assert_close(out, ref, 1e-5, "out")
    assert_close(eg.grad, er.grad, 1e-4, "d embed")
    assert_close(peg.grad, per.grad, 1e-4, "d embed_pos")
    assert_close(xg.grad, xr.grad, 1e-4, "d x")
    for k, v in dec.named_parameters():
        assert_close(v.grad, wr[k].grad, 1e-4, "grad " + k)


# ---------------------------------------------------------------------------------- scene
def make_scene(g, cfg, dev):
    m = JointEncoding(cfg, T(g["bound"]), T(g["half_len"])).to(dev)
    m.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    return m


def cfg_for(name):
    cfg = synth.config_plumbing()
    if name == "scene_s75.npz":
        cfg["training"].update(n_samples_d=50, n_range_d=25, n_samples=75)
    if name == "scene_nd0.npz":         # scene_rep.py:166-167: no uniform samples, z_vals = the depth-guided ones
        cfg["training"].update(n_samples_d=0, n_range_d=16, n_samples=16)
    return cfg


@pytest.mark.parametrize("name", ["scene_cfg1.npz", "scene_s75.npz", "scene_nd0.npz"])
def test_scene_eval_golden(dev, name):
    g = load_golden(name)
    m = make_scene(g, cfg_for(name), dev).eval()
    ro, rd, td = (T(g[k]).to(dev) for k in ("rays_o", "rays_d", "target_d"))
    with torch.no_grad():
        out = m.forward(ro, rd, None, td, noise=T(g["noise"]).to(dev))
    assert np.array_equal(out["z_vals"].cpu().numpy(), g["eval.z_vals"]), "sample placement must be bit-exact"
    for k in ("raw", "rgb", "depth", "disp_map", "acc_map", "depth_var"):
        assert_close(out[k], g["eval." + k], 1e-4, "eval " + k)
    with torch.no_grad():
        nd = m.render_rays(ro, rd, target_d=None, noise=T(g["noise_nodepth"]).to(dev))
    assert np.array_equal(nd["z_vals"].cpu().numpy(), g["nodepth.z_vals"])
    assert_close(nd["raw"], g["nodepth.raw"], 1e-4, "no-depth raw")
    assert_close(nd["depth"], g["nodepth.depth"], 1e-4, "no-depth depth")


def assert_rays_close(a, b, tol, what, max_flipped_rays):
    """Per-ray gradients: a ray's gradient changes by O(1) when ONE of its ~5000 discrete decisions (a ReLU, the first
    sign change of the SDF along the ray) sits within rounding noise of its threshold and falls the other way.  The
    fp32-MFMA path reproduces the reference's decisions on these fixtures (max_flipped_rays = 0); the f16x3 path carries
    operands with 23 instead of 24 significant bits and is allowed that many rays with a flipped decision -- every other
    ray must agree to `tol` of the maximum."""
    a = a.detach().cpu().numpy()
    b = np.asarray(b)
    per_ray = np.abs(a - b).max(1) / (np.abs(b).max() + 1e-30)
    off = np.nonzero(per_ray > tol)[0]
    print(f"  {what}: {len(off)} of {len(per_ray)} rays off by > {tol:g} of max (allowed {max_flipped_rays}); "
          f"median ray error {np.median(per_ray):.2e}")
    assert len(off) <= max_flipped_rays, f"{what}: rays {off.tolist()} differ, worst {per_ray.max():.3e}"


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16x6"])
@pytest.mark.parametrize("name", ["scene_cfg1.npz", "scene_s75.npz", "scene_nd0.npz"])
@pytest.mark.parametrize("tag,emd", [("emd", 0.01), ("noemd", 0.0)])
def test_scene_train_golden(dev, name, tag, emd, precision):
    """The reference's own training step on its own fixtures.  "f32" (fp32-input MFMA) and "bf16x6" (fp32 operands carried
    exactly as three bf16 pieces, six products) are held to the SAME gates: no ray with a flipped decision, every
    gradient within 5e-4 of its maximum; "f16x3" (22-23 operand bits) is allowed one flipped ray."""
    g = load_golden(name)
    cfg = cfg_for(name)
    m = make_scene(g, cfg, dev).train()
    assert m.decoder_precision == "bf16x6", "fp32 operands carried exactly (three bf16 pieces) are the default training path"
    m.decoder_precision = precision
    ro = T(g["rays_o"]).to(dev).requires_grad_(True)
    rd = T(g["rays_d"]).to(dev).requires_grad_(True)
    ret = m.forward(ro, rd, T(g["target_rgb"]).to(dev), T(g["target_d"]).to(dev), EMD_w=emd,
                    noise=T(g["noise"]).to(dev))
    for k in ("rgb", "depth", "rgb_loss", "depth_loss", "sdf_loss", "fs_loss", "psnr"):
        assert_close(ret[k].reshape(-1), g[f"{tag}.{k}"].reshape(-1), 1e-4, k)
    tr = cfg["training"]
    loss = (tr["rgb_weight"] * ret["rgb_loss"] + tr["depth_weight"] * ret["depth_loss"]
            + tr["sdf_weight"] * ret["sdf_loss"] + tr["fs_weight"] * ret["fs_loss"])
    assert_close(loss, g[f"{tag}.loss"], 1e-4, "total loss")
    loss.backward()
    exact_operands = precision in ("f32", "bf16x6")
    flips = 0 if exact_operands else 1
    assert_rays_close(ro.grad, g[f"{tag}.d_rays_o"], 5e-4, "d rays_o", flips)
    assert_rays_close(rd.grad, g[f"{tag}.d_rays_d"], 5e-4, "d rays_d", flips)
    for k, v in m.named_parameters():
        if v.numel():
            if exact_operands:
                assert_close(v.grad, g[f"{tag}.g.{k}"], 5e-4, "grad " + k)
            else:       # a flipped decision moves a handful of the entries that ray touches (one ReLU unit = one bias entry)
                assert_grad_close(v.grad, g[f"{tag}.g.{k}"], 5e-4, "grad " + k,
                                  outlier_frac=max(2e-3, 1.0 / v.numel()))


def test_scene_module_api_and_queries(dev):
    """query_* take pre-normalised coords, run_network takes local coords; deepcopy / state_dict round trip."""
    import copy
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    m = make_scene(g, cfg, dev).eval()
    cpu = path_cpu.CpuScene(cfg, g["bound"], g["half_len"])
    cpu.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    torch.manual_seed(9)
    pts = torch.rand(500, 3) * 1.6 - 0.8
    with torch.no_grad():
        ref = cpu.run_network(pts.reshape(20, 25, 3))
        out = m.run_network(pts.reshape(20, 25, 3).to(dev))
        assert out.shape == (20, 25, 10)
        assert_close(out, ref, 1e-4, "run_network")
        xn = (pts.double() + 1) / 2
        refq = cpu.query_normalised(xn)
        assert_close(m.query_color_sdf(xn.to(dev)[:, None, :]), refq, 1e-4, "query_color_sdf")
        assert_close(m.query_sdf(xn.to(dev)), refq[:, 3:4], 1e-4, "query_sdf")
        assert_close(m.query_color(xn.to(dev)), torch.sigmoid(refq[:, :3]), 1e-4, "query_color")
        assert m.query_sdf_entropy_prob(xn.to(dev)).shape == (500, 7)
        m2 = copy.deepcopy(m)
        assert_close(m2.run_network(pts.to(dev)), out.reshape(-1, 10), 1e-7, "deepcopy")
        m3 = JointEncoding(cfg, T(g["bound"]), T(g["half_len"])).to(dev)
        m3.load_state_dict(m.state_dict())
        assert torch.equal(m3.run_network(pts.to(dev)), m.run_network(pts.to(dev)))


def test_default_noise_draw_keeps_cpu_rng_stream(dev):
    """Without an explicit noise tensor the module must consume torch's CPU generator exactly like
    scene_rep.py:176 (torch.rand(N,S)), so later pixel-sampling calls stay bit-identical."""
    g = load_golden("scene_cfg1.npz")
    m = make_scene(g, cfg_for("scene_cfg1.npz"), dev).eval()
    ro, rd, td = (T(g[k]).to(dev) for k in ("rays_o", "rays_d", "target_d"))
    torch.manual_seed(77)
    expect_noise = torch.rand(256, 16)
    expect_next = torch.randn(5)
    torch.manual_seed(77)
    with torch.no_grad():
        out = m.forward(ro, rd, None, td)
    assert torch.equal(torch.randn(5), expect_next)
    with torch.no_grad():
        out2 = m.forward(ro, rd, None, td, noise=expect_noise.to(dev))
    assert torch.equal(out["z_vals"], out2["z_vals"])


def test_frozen_map_skips_param_grads_but_keeps_pose_grads(dev):
    """Tracking (mipsfusion.py:470-577) optimises the pose only.  With requires_grad False on the map the wgrad and
    scatter kernels are skipped; d(rays) must be unchanged and no parameter gradient may appear."""
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    m = make_scene(g, cfg, dev).train()
    for prm in m.parameters():
        prm.requires_grad_(False)
    ro = T(g["rays_o"]).to(dev).requires_grad_(True)
    rd = T(g["rays_d"]).to(dev).requires_grad_(True)
    ret = m.forward(ro, rd, T(g["target_rgb"]).to(dev), T(g["target_d"]).to(dev), EMD_w=0.0, noise=T(g["noise"]).to(dev))
    path_cpu.total_loss(ret, cfg["training"]).backward()
    assert_close(ro.grad, g["noemd.d_rays_o"], 5e-4, "d rays_o (frozen map)")
    assert_close(rd.grad, g["noemd.d_rays_d"], 5e-4, "d rays_d (frozen map)")
    assert all(prm.grad is None for prm in m.parameters())


def test_random_optimizer_fitness_slice(dev):
    """a12: RandomOptimizer.get_fitness (RandomOptimizer.py:113-131): P particles x n surface points, forward-only
    run_network, masked mean |sdf * trunc| * 1000 -- against the oracle."""
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    m = make_scene(g, cfg, dev).eval()
    cpu = path_cpu.CpuScene(cfg, g["bound"], g["half_len"])
    cpu.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    torch.manual_seed(3)
    P, n = 37, 100                                   # ragged on purpose
    world = torch.rand(P, n, 3) * 1.6 - 0.8
    td = torch.rand(n, 1) * 2
    td[::9] = 0.0                                    # invalid depth pixels are masked out
    trunc = cfg["training"]["trunc"]
    with torch.no_grad():
        ref_sdf = cpu.run_network(world)[..., 3:4].squeeze(-1) * trunc
        valid = (td > 0).float().squeeze(-1)[None]
        ref = torch.mean(valid * ref_sdf.abs(), dim=-1)
        got = ops.ro_fitness(m.run_network(world.to(dev)), td.squeeze(-1).to(dev), trunc)
    assert_close(got * 1000.0, ref * 1000.0, 1e-4, "fitness")


def test_centre_length_normalisation_and_hash16_vs_oracle(dev):
    """FastCaMo-large style submap (BASELINE config 4): hash 2^16, use_bound_normalize False -> (x + L) / 2L
    (scene_rep.py:142), one training iteration against the oracle."""
    cfg = synth.config_large_submap()
    cfg["training"].update(n_samples_d=11, n_range_d=5, n_samples=16)
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    torch.manual_seed(5)
    m = JointEncoding(cfg, bb, nf).to(dev).train()
    with torch.no_grad():
        m.embed_fn.params.copy_((torch.randn(m.embed_fn.params.shape) * 0.2).to(dev))
    cpu = path_cpu.CpuScene(cfg, bb, nf)
    cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    frame = synth.make_frame(cfg, seed=2)
    H, W = frame["depth"].shape
    idx = torch.randperm(H * W)[:333]                # ragged ray count
    ro, rd, rgb, d = synth.ray_batch(frame, idx, frame["c2w"])
    noise = torch.rand(333, 16)
    ret = m.forward(ro.to(dev), rd.to(dev), rgb.to(dev), d.to(dev), noise=noise.to(dev))
    ref = cpu.train_forward(ro, rd, rgb, d, noise, 0.01)
    for k in ("rgb", "depth", "rgb_loss", "depth_loss", "sdf_loss", "fs_loss"):
        assert_close(ret[k], ref[k], 1e-4, k)
    path_cpu.total_loss(ret, cfg["training"]).backward()
    path_cpu.total_loss(ref, cfg["training"]).backward()
    assert_grad_close(m.embed_fn.params.grad, cpu.embed_fn.params.grad, 5e-4, "grid gradient (hash 2^16)")


@pytest.mark.parametrize("n_rays", [1, 3, 65])
def test_ragged_ray_counts_and_empty_batch(dev, n_rays):
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    m = make_scene(g, cfg, dev).train()
    cpu = path_cpu.CpuScene(cfg, g["bound"], g["half_len"])
    cpu.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    sl = slice(0, n_rays)
    args = [T(g[k])[sl] for k in ("rays_o", "rays_d", "target_rgb", "target_d")]
    noise = T(g["noise"])[sl]
    ret = m.forward(*[a.to(dev) for a in args], noise=noise.to(dev))
    ref = cpu.train_forward(*args, noise, 0.01)
    for k in ("rgb", "depth", "rgb_loss", "sdf_loss", "fs_loss"):
        assert_close(ret[k], ref[k], 1e-4, f"{k} (N={n_rays})")
    path_cpu.total_loss(ret, cfg["training"]).backward()
    assert torch.isfinite(m.embed_fn.params.grad).all()
    # empty batch: every entry point returns without launching
    empty = [a[:0].to(dev) for a in args]
    m.eval()
    with torch.no_grad():
        out = m.forward(empty[0], empty[1], None, empty[3], noise=noise[:0].to(dev))
    assert out["rgb"].shape == (0, 3) and out["raw"].shape == (0, 16, 10)


def test_maximum_samples_per_ray(dev):
    """S = 256 is the most a ray kernel accepts (4 samples per lane); S = 257 is refused with an error."""
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    cfg["training"].update(n_samples_d=200, n_range_d=56, n_samples=256)
    m = make_scene(g, cfg, dev).eval()
    cpu = path_cpu.CpuScene(cfg, g["bound"], g["half_len"])
    cpu.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    ro, rd, td = (T(g[k])[:40] for k in ("rays_o", "rays_d", "target_d"))
    noise = torch.rand(40, 256)
    with torch.no_grad():
        out = m.forward(ro.to(dev), rd.to(dev), None, td.to(dev), noise=noise.to(dev))
        ref = cpu.render_rays(ro, rd, td, noise)
    assert np.array_equal(out["z_vals"].cpu().numpy(), ref["z_vals"].numpy())
    assert_close(out["depth"], ref["depth"], 1e-4, "depth at S=256")
    cfg["training"].update(n_samples_d=201)
    m2 = make_scene(g, cfg, dev).eval()
    with pytest.raises(RuntimeError, match="samples per ray"):
        m2.forward(ro.to(dev), rd.to(dev), None, td.to(dev), noise=torch.rand(40, 257, device=dev))


def test_no_valid_depth_gives_nan_losses_like_reference(dev):
    g = load_golden("scene_cfg1.npz")
    m = make_scene(g, cfg_for("scene_cfg1.npz"), dev).train()
    ro, rd = T(g["rays_o"]).to(dev), T(g["rays_d"]).to(dev)
    ret = m.forward(ro, rd, T(g["target_rgb"]).to(dev), torch.zeros(256, 1, device=dev), noise=T(g["noise"]).to(dev))
    assert math.isnan(float(ret["depth_loss"])) and math.isnan(float(ret["fs_loss"])) and math.isnan(float(ret["sdf_loss"]))
    assert float(ret["rgb_loss"]) == 0.0            # rgb_missing = 0 masks every ray


# ----------------------------------------------------------------------------------- Adam
def test_fused_adam_golden(dev):
    g = load_golden("adam.npz")
    grid = torch.nn.Parameter(T(g["grid0"]).to(dev))
    dec = torch.nn.Parameter(T(g["dec0"]).to(dev))
    opt = FusedAdam([{"params": [dec], "weight_decay": 1e-6, "lr": 0.01}, {"params": [grid], "eps": 1e-15, "lr": 0.01}],
                    betas=(0.9, 0.99))
    for s in range(g["grid_grads"].shape[0]):
        grid.grad, dec.grad = T(g["grid_grads"][s]).to(dev), T(g["dec_grads"][s]).to(dev)
        opt.step()
        np.testing.assert_allclose(grid.detach().cpu().numpy(), g["grid_traj"][s], rtol=2e-5, atol=1e-8)
        np.testing.assert_allclose(dec.detach().cpu().numpy(), g["dec_traj"][s], rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(opt.state[grid]["exp_avg"].cpu().numpy(), g["grid_m"], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(opt.state[grid]["exp_avg_sq"].cpu().numpy(), g["grid_v"], rtol=1e-5, atol=1e-13)
    opt.step(zero_grad=True)
    assert float(grid.grad.abs().max()) == 0.0 and float(dec.grad.abs().max()) == 0.0


def test_fused_adam_odd_sizes_vs_torch(dev):
    torch.manual_seed(1)
    for n in (1, 3, 5, 1027):
        p0, gr = torch.randn(n), torch.randn(n)
        a = torch.nn.Parameter(p0.clone().to(dev))
        b = torch.nn.Parameter(p0.clone())
        oa = FusedAdam([a], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
        ob = torch.optim.Adam([b], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
        for _ in range(3):
            a.grad, b.grad = gr.clone().to(dev), gr.clone()
            oa.step(), ob.step()
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=1e-8)


def test_fused_adam_multi_tensor_group_vs_torch(dev):
    """The decoder's ten small tensors go through ONE launch; compare with torch.optim.Adam (weight decay on)."""
    torch.manual_seed(2)
    shapes = [(128, 51), (128,), (128, 128), (128,), (3, 115), (3,), (128, 96), (128,), (5, 128), (5,)]
    ps = [torch.randn(*s) * 0.1 for s in shapes]
    a = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    b = [torch.nn.Parameter(p.clone()) for p in ps]
    oa = FusedAdam([{"params": a, "weight_decay": 1e-6, "lr": 0.01}], betas=(0.9, 0.99))
    ob = torch.optim.Adam([{"params": b, "weight_decay": 1e-6, "lr": 0.01}], betas=(0.9, 0.99))
    for _ in range(4):
        for x, y in zip(a, b):
            g = torch.randn_like(y) * 0.01
            x.grad, y.grad = g.clone().to(dev), g
        oa.step(), ob.step()
    for x, y in zip(a, b):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().numpy(), rtol=2e-5, atol=1e-8)


def test_fused_adam_small_groups_one_launch(dev):
    """Optimisers of tiny tensors only (the pose optimisers) take the one-launch path (device step counters, bias
    corrections and all updates in one workgroup): two groups with their own lr / eps / weight decay against
    torch.optim.Adam, and bit for bit against the multi-launch path (capturable=False)."""
    torch.manual_seed(5)
    r0, t0 = torch.randn(9, 4), torch.randn(9, 3)
    mk = lambda dev_: (torch.nn.Parameter(r0.clone().to(dev_)), torch.nn.Parameter(t0.clone().to(dev_)))
    (ra, ta), (rb, tb), (rc_, tc) = mk(dev), mk(dev), mk("cpu")
    groups = lambda r, t: [{"params": r, "lr": 1e-3}, {"params": t, "lr": 1e-2, "eps": 1e-10, "weight_decay": 1e-4}]
    oa = FusedAdam(groups(ra, ta), betas=(0.9, 0.99), capturable=True)
    ob = FusedAdam(groups(rb, tb), betas=(0.9, 0.99), capturable=False)
    oc = torch.optim.Adam(groups(rc_, tc), betas=(0.9, 0.99))
    for _ in range(6):
        gr, gt = torch.randn(9, 4), torch.randn(9, 3)
        ra.grad, ta.grad = gr.clone().to(dev), gt.clone().to(dev)
        rb.grad, tb.grad = gr.clone().to(dev), gt.clone().to(dev)
        rc_.grad, tc.grad = gr.clone(), gt.clone()
        oa.step(zero_grad=True), ob.step(), oc.step()
        assert float(ra.grad.abs().max()) == 0.0 and float(ta.grad.abs().max()) == 0.0       # fused zero_grad
    assert torch.equal(ra, rb) and torch.equal(ta, tb)
    np.testing.assert_allclose(ra.detach().cpu().numpy(), rc_.detach().numpy(), rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(ta.detach().cpu().numpy(), tc.detach().numpy(), rtol=2e-5, atol=1e-8)


def test_fused_adam_whole_step_in_one_launch(dev):
    """capturable FusedAdam over tensors of ANY size takes mipsf_adam_step_all (one launch: every workgroup derives its
    group's bias corrections from the device step counter or takes the pair the previous step's last workgroup left, a
    two-level ticket finds the workgroup after which all have read; csrc/elementwise.hip): a large tensor (many
    workgroups, all eight XCDs) + odd-sized small ones in two groups against torch.optim.Adam and against the multi-launch
    path, over steps that include a learning-rate change (the left-behind scalars are tagged with step AND rate), a
    reset() and a missing gradient round trip; the ticket block ends every step at zero."""
    torch.manual_seed(8)
    big0, w0, b0 = torch.randn(3_000_017) * 0.1, torch.randn(37, 5), torch.randn(3)
    def mk(d):
        return [torch.nn.Parameter(t.clone().to(d)) for t in (big0, w0, b0)]
    pa, pb, pc = mk(dev), mk(dev), mk("cpu")
    groups = lambda ps: [{"params": ps[1:], "weight_decay": 1e-6, "lr": 0.01}, {"params": ps[:1], "eps": 1e-15, "lr": 0.01}]   # noqa: E731
    oa = FusedAdam(groups(pa), betas=(0.9, 0.99), capturable=True)
    ob = FusedAdam(groups(pb), betas=(0.9, 0.99), capturable=False)
    oc = torch.optim.Adam(groups(pc), betas=(0.9, 0.99))
    launches = {"n": 0}
    orig = ops.adam_step_all

    def counted(*a, **k):
        launches["n"] += 1
        return orig(*a, **k)
    ops.adam_step_all = counted
    try:
        for it in range(9):
            if it == 4:                                   # a schedule changes the rate between two steps
                for o in (oa, ob, oc):
                    for gq in o.param_groups:
                        gq["lr"] = 0.003
            if it == 6:                                   # fresh optimiser state, in place
                oa.reset(), ob.reset()
                oc = torch.optim.Adam(groups(pc), betas=(0.9, 0.99))
                for gq in oc.param_groups:
                    gq["lr"] = 0.003
            gs = [torch.randn_like(t) * (0.0 if (it == 2 and k == 0) else 1.0) for k, t in enumerate((big0, w0, b0))]
            gs[0][::3] = 0.0                              # dense semantics: entries without a gradient still move
            for ps, d in ((pa, dev), (pb, dev), (pc, "cpu")):
                for p_, g_ in zip(ps, gs):
                    p_.grad = g_.clone().to(d)
            oa.step(zero_grad=True), ob.step(), oc.step()
            assert int(oa._ticket[:528].abs().sum()) == 0, "tickets must end every step at zero"
            assert all(float(p_.grad.abs().max()) == 0.0 for p_ in pa)
    finally:
        ops.adam_step_all = orig
    assert launches["n"] == 9, "the capturable optimiser did not take the one-launch path"
    for a, b, c in zip(pa, pb, pc):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=3e-7, atol=1e-9)
        np.testing.assert_allclose(a.detach().cpu().numpy(), c.detach().numpy(), rtol=2e-5, atol=1e-8)


def test_render_losses_finished_in_the_render_launch(dev):
    """mipsf_render_fwd with a ticket (the last workgroup of the render kernel finishes the losses, one launch) against the
    two-launch form mipsf_render_fwd: every output and all eight loss entries, at ray counts that leave the last
    16-ray workgroup partly empty and that need one / several workgroups; the ticket is left at zero and repeated calls
    agree bit for bit."""
    import ctypes as C
    from mipsfusion_amd._lib import dptr, lib, stream_ptr
    torch.manual_seed(12)
    cfg = synth.config_headline()
    S = 64
    rc = ops.make_render_cfg(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"], 43, 21, 0.01)
    for N in (1, 15, 16, 17, 1000, 4096):
        raw = torch.randn(N, S, 10, device=dev)
        raw[..., 3] = torch.linspace(1.0, -1.0, S, device=dev)[None] + 0.1 * torch.randn(N, S, device=dev)
        z = torch.sort(torch.rand(N, S, device=dev) * 4.0 + 0.1, dim=1).values
        t_rgb, t_d = torch.rand(N, 3, device=dev), torch.rand(N, 1, device=dev) * 4.0
        t_d[::7] = 0.0
        counts = torch.randint(0, S, (N, 2), dtype=torch.int32, device=dev)
        lw = torch.tensor([1.0, 0.0, 1000.0, 10.0], device=dev)

        def run(ticket):
            f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)      # noqa: E731
            outs = [f(N, 3), f(N), f(N), f(N), f(N)]
            losses, partial, total = f(8), f(N * 8), f(1)
            a = _lib.RenderFwdArgs.new(N=N, S=S, raw=dptr(raw), z_vals=dptr(z), target_rgb=dptr(t_rgb), target_d=dptr(t_d),
                                       counts=dptr(counts, torch.int32), cfg=C.pointer(rc), rgb=dptr(outs[0]), depth=dptr(outs[1]),
                                       depth_var=dptr(outs[2]), disp=dptr(outs[3]), acc=dptr(outs[4]), losses=dptr(losses),
                                       partial=dptr(partial), loss_weights=dptr(lw), loss_total=dptr(total),
                                       ticket=dptr(ticket, torch.int32) if ticket is not None else None)
            assert lib().mipsf_render_fwd(C.byref(a), stream_ptr()) == 0
            return outs, losses, total
        ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        o2, l2, t2 = run(None)
        for rep in range(3):
            o1, l1, t1 = run(ticket)
            assert int(ticket[0]) == 0
            for a, b in zip(o1, o2):
                assert torch.equal(a, b)
            # fp64 sums of the same fp32 rows in another (fixed) order: equal up to the last fp32 bit of the quotient
            np.testing.assert_allclose(l1.cpu().numpy(), l2.cpu().numpy(), rtol=3e-7, equal_nan=True)
            np.testing.assert_allclose(t1.cpu().numpy(), t2.cpu().numpy(), rtol=3e-7, equal_nan=True)
            if rep:
                assert torch.equal(l1, l_prev) or (torch.isnan(l1) == torch.isnan(l_prev)).all()
            l_prev = l1


def test_render_backward_written_by_the_forward_launch(dev):
    """mipsf_render_fwd with `draw` (S <= 128: the ray staged once, the objective's gradient written by the same launch) against
    the separate kernels: per-ray outputs equal the two-launch forward's bit for bit, `draw` equals mipsf_render_bwd's for
    g_total = 1 bit for bit (EMD on and off, rays without depth, partly empty workgroups, one and two samples per lane);
    KEEP_IF_UNIT leaves the buffer alone for g_total = 1 and rewrites it -- bit-equal to the plain backward -- otherwise; through
    autograd every way of calling backward gives the gradients of the unfused path."""
    import ctypes as C
    from mipsfusion_amd._lib import dptr, lib, stream_ptr
    from mipsfusion_amd.helper_functions.utils import backward_from_one
    from mipsfusion_amd.model.scene_rep import _RenderFn
    torch.manual_seed(21)
    cfg = synth.config_headline()
    lw = torch.tensor([1.0, 0.1, 1000.0, 10.0], device=dev)
    one, two = torch.ones(1, device=dev), torch.full((1,), 2.0, device=dev)
    for S, N, emd in ((16, 17, 0.01), (64, 1000, 0.01), (64, 4096, 0.0), (75, 333, 0.01), (128, 50, 0.01), (100, 16, 0.0)):
        rc = ops.make_render_cfg(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"], S - S // 3, S // 3, emd)
        raw = torch.randn(N, S, 10, device=dev)
        raw[..., 3] = torch.linspace(1.0, -1.0, S, device=dev)[None] + 0.1 * torch.randn(N, S, device=dev)
        raw[..., 5:] = torch.softmax(raw[..., 5:], -1)
        raw[::5, :, 3] = raw[::5, :, 3].abs()                    # rays without a sign change
        z = torch.sort(torch.rand(N, S, device=dev) * 4.0 + 0.1, dim=1).values
        t_rgb, t_d = torch.rand(N, 3, device=dev), torch.rand(N, 1, device=dev) * 4.0
        t_d[::7] = 0.0
        counts = torch.randint(0, S, (N, 2), dtype=torch.int32, device=dev)
        res = ops.render_fwd(raw, z, t_rgb, t_d, counts, rc, N, S, True, want_weights=True, loss_weights=lw, want_draw=True)
        rgb, depth, var, disp, acc, weights, losses, total, draw = res
        # the two-launch forward (the component-by-component kernel)
        f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)      # noqa: E731
        o2 = [f(N, 3), f(N), f(N), f(N), f(N), f(N, S)]
        l2, partial, t2 = f(8), f(N * 8), f(1)
        a = _lib.RenderFwdArgs.new(N=N, S=S, raw=dptr(raw), z_vals=dptr(z), target_rgb=dptr(t_rgb), target_d=dptr(t_d),
                                   counts=dptr(counts, torch.int32), cfg=C.pointer(rc), rgb=dptr(o2[0]), depth=dptr(o2[1]),
                                   depth_var=dptr(o2[2]), disp=dptr(o2[3]), acc=dptr(o2[4]), weights=dptr(o2[5]), losses=dptr(l2),
                                   partial=dptr(partial), loss_weights=dptr(lw), loss_total=dptr(t2))
        assert lib().mipsf_render_fwd(C.byref(a), stream_ptr()) == 0
        for x, y in zip((rgb, depth, var, disp, acc, weights), o2):
            assert torch.equal(x, y), (S, N)
        np.testing.assert_allclose(losses.cpu().numpy(), l2.cpu().numpy(), rtol=3e-7, equal_nan=True)
        np.testing.assert_allclose(total.cpu().numpy(), t2.cpu().numpy(), rtol=3e-7, equal_nan=True)
        # the gradient
        ref1 = ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=one, loss_weights=lw)
        assert torch.equal(draw, ref1), (S, N, float((draw - ref1).abs().max()))
        ref2 = ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=two, loss_weights=lw)
        mark = torch.full_like(raw, 7.0)
        kept = ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=one, loss_weights=lw,
                              keep_draw=mark)
        assert kept is mark and bool((mark == 7.0).all())
        again = ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=two, loss_weights=lw,
                               keep_draw=mark)
        assert torch.equal(again, ref2)
    # edges: one and two samples per ray, one ray, and a batch without a single valid depth (depth_loss = 0 / 0, fs / sdf weights
    # from zero counts: NaN where the separate kernels have NaN, bit for bit)
    same = lambda a_, b_: torch.equal(a_.view(torch.int32), b_.view(torch.int32))      # noqa: E731
    for S, N, no_depth in ((1, 5, False), (2, 1, False), (64, 33, True), (75, 16, True)):
        rc = ops.make_render_cfg(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"], S, 0, 0.01)
        raw = torch.randn(N, S, 10, device=dev)
        z = torch.sort(torch.rand(N, S, device=dev) * 4.0 + 0.1, dim=1).values
        t_rgb = torch.rand(N, 3, device=dev)
        t_d = torch.zeros(N, 1, device=dev) if no_depth else torch.rand(N, 1, device=dev) * 4.0
        counts = torch.zeros(N, 2, dtype=torch.int32, device=dev) if no_depth else torch.randint(0, S + 1, (N, 2), dtype=torch.int32, device=dev)
        res = ops.render_fwd(raw, z, t_rgb, t_d, counts, rc, N, S, True, loss_weights=lw, want_draw=True)
        losses, draw = res[6], res[8]
        ref = ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=one, loss_weights=lw)
        assert same(draw, ref), (S, N, no_depth)
        if no_depth:
            assert bool(torch.isnan(losses[1]))
        nan = torch.full((1,), float("nan"), device=dev)
        again = ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=nan, loss_weights=lw,
                               keep_draw=draw.clone())
        assert same(again, ops.render_bwd(raw, z, t_rgb, t_d, counts, losses, rc, None, None, None, N, S, g_total=nan, loss_weights=lw))
    with pytest.raises(ValueError):
        ops.render_fwd(torch.randn(4, 129, 10, device=dev), torch.rand(4, 129, device=dev), torch.rand(4, 3, device=dev),
                       torch.rand(4, 1, device=dev), torch.zeros(4, 2, dtype=torch.int32, device=dev), rc, 4, 129, True,
                       loss_weights=lw, want_draw=True)

    # ---- through autograd
    S, N = 64, 500
    rc = ops.make_render_cfg(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"], 43, 21, 0.01)
    raw0 = torch.randn(N, S, 10, device=dev)
    raw0[..., 3] = torch.linspace(1.0, -1.0, S, device=dev)[None] + 0.1 * torch.randn(N, S, device=dev)
    z = torch.sort(torch.rand(N, S, device=dev) * 4.0 + 0.1, dim=1).values
    t_rgb, t_d = torch.rand(N, 3, device=dev), torch.rand(N, 1, device=dev) * 4.0
    counts = torch.randint(0, S, (N, 2), dtype=torch.int32, device=dev)

    def grads(fuse, how):
        _RenderFn.fuse_backward = fuse
        try:
            raw = raw0.clone().requires_grad_(True)
            res = _RenderFn.apply(raw, z, t_rgb, t_d, counts, rc, N, S, True, lw, None)
            total, rgb = res[6], res[0]
            if how == "unit":
                backward_from_one(total)
            elif how == "plain":
                total.backward()
            elif how == "scaled":
                (total * 2.0).backward()
            elif how == "with_rgb":
                (total + rgb.sum()).backward()
            elif how == "twice":
                backward_from_one(total, retain_graph=True)
                (total * 3.0).backward(retain_graph=True)
                backward_from_one(total)
            return raw.grad.clone()
        finally:
            _RenderFn.fuse_backward = True
    for how in ("unit", "plain", "scaled", "with_rgb", "twice"):
        assert torch.equal(grads(True, how), grads(False, how)), how


def test_fused_adam_reset_equals_fresh_optimizer(dev):
    """FusedAdam.reset() (in place, also under capturable=True) must continue exactly like a newly built
    torch.optim.Adam -- the reference rebuilds its pose optimiser every frame (mipsfusion.py:472-475)."""
    torch.manual_seed(4)
    for capturable in (False, True):
        p0 = torch.randn(7, 4)
        a, b = torch.nn.Parameter(p0.clone().to(dev)), torch.nn.Parameter(p0.clone())
        oa = FusedAdam([{"params": a, "lr": 0.01}], capturable=capturable)
        moments = None
        for rnd in range(3):
            ob = torch.optim.Adam([{"params": b, "lr": 0.01}])          # fresh every round
            if rnd:
                oa.reset()
                assert oa.state[a]["exp_avg"] is moments                 # cleared, not replaced
            for _ in range(4):
                g = torch.randn(7, 4)
                a.grad, b.grad = g.clone().to(dev), g
                oa.step(), ob.step()
            moments = oa.state[a]["exp_avg"]
            np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().numpy(), rtol=2e-5, atol=1e-8)


def test_pose_rays_matches_torch_composition(dev):
    """Fused ray building == qt_to_transform_matrix + gather + sum(d_cam * R, -1) (mipsfusion.py:320-322), values
    and gradients wrt quaternion / translation; includes python-style negative owner indices and mixed waves."""
    from mipsfusion_amd.helper_functions.geometry_helper import qt_to_transform_matrix
    g = load_golden("quaternion.npz")
    torch.manual_seed(4)
    N, F, K = 1000, 2, 9
    rot0, trans0 = T(g["rot"]).to(dev), T(g["trans"]).to(dev)
    fixed = qt_to_transform_matrix(torch.randn(F, 4, device=dev), torch.randn(F, 3, device=dev))
    owner = torch.randint(-(F + K), F + K, (N,), device=dev)
    owner[:300] = 3                                    # uniform waves take the shuffle path
    d_cam = torch.randn(N, 3, device=dev)
    go, gd = torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)
    rot_a, trans_a = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
    ro, rd = ops.pose_rays(rot_a, trans_a, fixed, owner, d_cam)
    (ro * go).sum().add((rd * gd).sum()).backward()
    rot_b, trans_b = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
    poses = torch.cat([fixed, qt_to_transform_matrix(rot_b, trans_b)], 0)
    rd_ref = torch.sum(d_cam[..., None, :] * poses[owner, :3, :3], -1)
    ro_ref = poses[owner, :3, -1]
    (ro_ref * go).sum().add((rd_ref * gd).sum()).backward()
    assert_close(ro, ro_ref, 1e-6, "rays_o")
    assert_close(rd, rd_ref, 1e-6, "rays_d")
    assert_close(rot_a.grad, rot_b.grad, 2e-5, "d quaternion")
    assert_close(trans_a.grad, trans_b.grad, 2e-5, "d translation")
    # the backward is ONE launch whose scratch carries a ticket it must leave at zero: repeated calls (same and other
    # sizes, several workgroups) have to keep giving the same gradients
    for _ in range(3):
        rot_r, trans_r = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
        ro2, rd2 = ops.pose_rays(rot_r, trans_r, fixed, owner, d_cam)
        (ro2 * go).sum().add((rd2 * gd).sum()).backward()
        assert_close(rot_r.grad, rot_b.grad, 2e-5, "d quaternion, repeated call")
        assert_close(trans_r.grad, trans_b.grad, 2e-5, "d translation, repeated call")
        rot_s, trans_s = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
        ro4, rd4 = ops.pose_rays(rot_s, trans_s, fixed, owner[:70], d_cam[:70])
        (ro4 * go[:70]).sum().add((rd4 * gd[:70]).sum()).backward()
        assert torch.isfinite(rot_s.grad).all()
    # opt-in: the backward adds straight into rot.grad / trans.grad (twice -> twice the gradient)
    rot_i, trans_i = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
    for _ in range(2):
        ro6, rd6 = ops.pose_rays(rot_i, trans_i, fixed, owner, d_cam, accumulate_in_place=True)
        (ro6 * go).sum().add((rd6 * gd).sum()).backward()
    assert_close(rot_i.grad, 2 * rot_b.grad, 2e-5, "d quaternion, accumulated in place twice")
    assert_close(trans_i.grad, 2 * trans_b.grad, 2e-5, "d translation, accumulated in place twice")
    # gather of the ray table's rows + ray construction in one launch == the two separate ops, forward and backward
    table = torch.randn(50, 60, 7, device=dev)
    rows_t = torch.randint(-3000, 3000, (N,), device=dev)
    rot_g, trans_g = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
    ro7, rd7, rgb7, dep7 = ops.gather_pose_rays(table, rows_t, rot_g, trans_g, fixed, owner)
    dc, rgb_ref, dep_ref = ops.gather_rays(table, rows_t, split=True)
    rot_h, trans_h = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
    ro8, rd8 = ops.pose_rays(rot_h, trans_h, fixed, owner, dc)
    assert torch.equal(rgb7, rgb_ref) and torch.equal(dep7, dep_ref)
    assert torch.equal(ro7, ro8) and torch.equal(rd7, rd8)
    (ro7 * go).sum().add((rd7 * gd).sum()).backward()
    (ro8 * go).sum().add((rd8 * gd).sum()).backward()
    assert_close(rot_g.grad, rot_h.grad, 1e-6, "d quaternion, fused gather + pose rays")
    assert_close(trans_g.grad, trans_h.grad, 1e-6, "d translation, fused gather + pose rays")
    # a batch of 40 000 rays: many workgroups (per-workgroup partial rows + ticket)
    Nb = 40000
    owner_b = torch.randint(0, F + K, (Nb,), device=dev)
    owner_b[:20000] = torch.arange(20000, device=dev) // 2000 % (F + K)         # long uniform stretches
    d_b, go_b, gd_b = torch.randn(Nb, 3, device=dev), torch.randn(Nb, 3, device=dev), torch.randn(Nb, 3, device=dev)
    for _ in range(2):
        rot_m, trans_m = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
        ro5, rd5 = ops.pose_rays(rot_m, trans_m, fixed, owner_b, d_b)
        (ro5 * go_b).sum().add((rd5 * gd_b).sum()).backward()
        rot_n, trans_n = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
        poses_n = torch.cat([fixed, qt_to_transform_matrix(rot_n, trans_n)], 0)
        ((poses_n[owner_b, :3, -1] * go_b).sum() + (torch.sum(d_b[..., None, :] * poses_n[owner_b, :3, :3], -1) * gd_b).sum()).backward()
        assert_close(rot_m.grad, rot_n.grad, 5e-5, "d quaternion, 40 000 rays")
        assert_close(trans_m.grad, trans_n.grad, 5e-5, "d translation, 40 000 rays")
    # and the chain alone against the reference's own autograd (golden)
    rot_c, trans_c = T(g["rot"]).to(dev).requires_grad_(True), T(g["trans"]).to(dev).requires_grad_(True)
    eye = torch.eye(3, device=dev)
    own = torch.arange(9, device=dev).repeat_interleave(3)
    ro3, rd3 = ops.pose_rays(rot_c, trans_c, None, own, eye.repeat(9, 1))      # rd rows = columns of R
    R = rd3.reshape(9, 3, 3).transpose(1, 2)
    assert_close(R, g["T"][:, :3, :3], 1e-6, "R from quaternion")
    gT = T(g["gT"]).to(dev)
    ((R * gT[:, :3, :3]).sum() + (ro3.reshape(9, 3, 3)[:, 0] * gT[:, :3, 3]).sum()).backward()
    assert_close(rot_c.grad, g["d_rot"], 2e-5, "d_rot vs reference autograd")
    assert_close(trans_c.grad, g["d_trans"], 2e-5, "d_trans vs reference autograd")


def test_forward_from_table_equals_gather_then_forward(dev):
    """JointEncoding.forward_from_table (row gather + ray construction + sample placement in one launch, ray + pose gradients
    in another) against ops.gather_pose_rays followed by forward(): sample placement and every output bit for bit, the
    pose gradients to fp32 rounding of their per-pose sums, the map gradients to the scatter's own reproducibility; ragged
    ray counts (last 16-ray workgroup of the backward partly empty), python-style negative owners, in-place accumulation."""
    from mipsfusion_amd.helper_functions.geometry_helper import qt_to_transform_matrix
    from mipsfusion_amd.helper_functions.utils import get_loss_from_ret
    torch.manual_seed(3)
    cfg = synth.config_headline()
    cfg["grid"]["hash_size"] = 14
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    m = JointEncoding(cfg, bb, nf).to(dev).train()
    frame = synth.make_frame(cfg, seed=5)
    table = torch.cat([frame["direction"], frame["rgb"], frame["depth"][..., None]], -1).reshape(-1, 7)[:50000].contiguous().to(dev)
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    F, K = 1, 5
    c2w = synth.default_pose(cfg)
    fixed = c2w[None].to(dev)
    rot0 = gh_quat(c2w[:3, :3])[None].repeat(K, 1).to(dev) + 0.01 * torch.randn(K, 4, device=dev)
    trans0 = c2w[None, :3, 3].repeat(K, 1).to(dev) + 0.02 * torch.randn(K, 3, device=dev)
    for N in (1, 17, 1000):
        rows = torch.randint(-50000, 50000, (N,), device=dev)
        owner = torch.randint(-(F + K), F + K, (N,), device=dev)
        noise = torch.rand(N, S, device=dev)
        outs = []
        for fused in (False, True):
            rot, trans = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            if fused:
                ret = m.forward_from_table(table, rows, rot, trans, fixed, owner, noise)
            else:
                rays_o, rays_d, rgb, depth = ops.gather_pose_rays(table, rows, rot, trans, fixed, owner)
                ret = m.forward(rays_o, rays_d, rgb, depth, noise=noise)
            loss = get_loss_from_ret(ret, cfg["training"])
            loss.backward()
            outs.append((ret, loss.detach(), rot.grad.clone(), trans.grad.clone(), m.embed_fn.params.grad.clone(),
                         m.decoder.pts_linear[0].weight.grad.clone()))
        (ra, la, gra, gta, gga, gwa), (rb, lb, grb, gtb, ggb, gwb) = outs
        for k in ("rgb", "depth", "rgb_loss", "depth_loss", "sdf_loss", "fs_loss"):
            assert torch.equal(ra[k], rb[k]) or (torch.isnan(ra[k]).all() and torch.isnan(rb[k]).all()), k
        assert torch.equal(la, lb) or (torch.isnan(la) and torch.isnan(lb))
        if torch.isfinite(la):
            assert_close(grb, gra, 2e-5, f"d quaternion, N={N}")
            assert_close(gtb, gta, 2e-5, f"d translation, N={N}")
            assert_close(ggb, gga, 1e-6, f"grid gradient, N={N}")
            assert_close(gwb, gwa, 1e-6, f"decoder gradient, N={N}")
    # in-place accumulation into the pose parameters' .grad (twice -> twice the gradient)
    rot, trans = rot0.clone().requires_grad_(True), trans0.clone().requires_grad_(True)
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        get_loss_from_ret(m.forward_from_table(table, rows, rot, trans, fixed, owner, noise, accumulate_in_place=True),
                          cfg["training"]).backward()
    assert_close(rot.grad, 2 * grb, 2e-5, "d quaternion accumulated in place twice")
    assert_close(trans.grad, 2 * gtb, 2e-5, "d translation accumulated in place twice")


def gh_quat(R):
    from mipsfusion_amd.helper_functions.geometry_helper import matrix_to_quaternion
    return matrix_to_quaternion(R[None])[0]


def test_pose_rays_bwd_ticket_reduction_under_load(dev):
    """The one-launch pose backward hands per-workgroup partial rows to the last workgroup through sc1 (write-through)
    stores, a `s_waitcnt vmcnt(0)` and a device-scope ticket -- no agent-scope fence (csrc/pose.hip).  Stress: 400
    workgroups (all eight XCDs), 2000 calls that ALTERNATE between two gradient sets while a second stream keeps the
    memory system unevenly busy; a row read stale would carry the other set's values.  Every word of every result must
    equal, bit for bit, what the same call gave on an idle device (the rows are summed in workgroup order), and that
    must agree with a two-pass reduction (fp64 index_add)."""
    import ctypes as C
    from mipsfusion_amd._lib import dptr, lib, stream_ptr
    torch.manual_seed(77)
    F, K, N = 1, 15, 256 * 400
    P = F + K
    rot = torch.randn(K, 4, device=dev)
    # one owner per wave, four different owners per workgroup: every LDS accumulator of a workgroup receives exactly one
    # add, so a call's result is a deterministic function of its inputs (mixed waves add with LDS float atomics, whose
    # order -- and rounding -- varies from run to run; they are covered by test_pose_rays_matches_torch_composition)
    owner = torch.arange(N, device=dev) // 64 % P
    d_cam = torch.randn(N, 3, device=dev)
    sets = [(torch.randn(N, 3, device=dev), torch.randn(N, 3, device=dev)) for _ in range(2)]
    scratch = torch.zeros(_lib.buffer_size(_lib.SIZE_POSE_RAYS_SCRATCH, N, F, K), device=dev)

    def call(k, out_rot, out_trans):
        go, gd = sets[k]
        rc = lib().mipsf_pose_rays_bwd(dptr(go), dptr(gd), dptr(rot), F, K, dptr(owner, torch.int64), dptr(d_cam),
                                          dptr(out_rot), dptr(out_trans), dptr(scratch), N, 0, stream_ptr())
        assert rc == 0
    want = []
    for k in range(2):
        r, t = torch.empty(K, 4, device=dev), torch.empty(K, 3, device=dev)
        call(k, r, t)
        torch.cuda.synchronize()
        # two-pass reference: per-pose sums in fp64, then the quaternion chain through torch autograd
        go, gd = sets[k]
        G = torch.zeros(P, 12, dtype=torch.float64, device=dev)
        G[:, :9].index_add_(0, owner, (gd.double()[:, :, None] * d_cam.double()[:, None, :]).reshape(N, 9))
        G[:, 9:].index_add_(0, owner, go.double())
        assert_close(t, G[F:, 9:], 2e-5, "d translation vs fp64 index_add")
        from mipsfusion_amd.helper_functions.geometry_helper import qt_to_transform_matrix
        rq = rot.double().clone().requires_grad_(True)
        R = qt_to_transform_matrix(rq, torch.zeros(K, 3, dtype=torch.float64, device=dev))[:, :3, :3]
        (R * G[F:, :9].reshape(K, 3, 3)).sum().backward()
        assert_close(r, rq.grad, 2e-5, "d quaternion vs fp64 two-pass reduction")
        want.append((r.clone(), t.clone()))
    n_rep = 2000
    res_r, res_t = torch.empty(n_rep, K, 4, device=dev), torch.empty(n_rep, K, 3, device=dev)
    busy = torch.empty(64 << 20, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):           # uneven background load: streaming fills of varying size
        for q in range(300):
            busy[: (1 + q % 7) << 22].add_(1.0)
    for q in range(n_rep):
        call(q & 1, res_r[q], res_t[q])
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(res_r[k::2], want[k][0].expand(n_rep // 2, K, 4)), "a stale partial row reached the reduction (d rot)"
        assert torch.equal(res_t[k::2], want[k][1].expand(n_rep // 2, K, 3)), "a stale partial row reached the reduction (d trans)"
    assert int(scratch[:1].view(torch.int32)[0]) == 0, "ticket left non-zero"


# ---------------------------------------------------------------------- optimisation loop
def test_ba_loop_trace_matches_reference(dev):
    """6 iterations of the mapping loop (mipsfusion.py:293-342) with pose + map optimisation: loss trace and final
    parameters against the reference run recorded in ba_trace.npz."""
    from mipsfusion_amd.helper_functions.geometry_helper import matrix_to_quaternion, qt_to_transform_matrix
    g = load_golden("ba_trace.npz")
    cfg = synth.config_plumbing()
    tr = cfg["training"]
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    m = JointEncoding(cfg, bb, nf).to(dev)
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("w0.")})
    m.train()
    frame = synth.make_frame(cfg, seed=int(g["frame_seed"]))
    H, W = frame["depth"].shape
    opt = FusedAdam([{"params": m.decoder.parameters(), "weight_decay": 1e-6, "lr": 0.01},
                     {"params": m.embed_fn.parameters(), "eps": 1e-15, "lr": 0.01}], betas=(0.9, 0.99))
    pose0 = T(g["pose0"]).to(dev)
    cur_trans = torch.nn.Parameter(pose0[:, :3, 3].clone())
    cur_rot = torch.nn.Parameter(matrix_to_quaternion(pose0[:, :3, :3]))
    pose_opt = torch.optim.Adam([{"params": cur_rot, "lr": 1e-3}, {"params": cur_trans, "lr": 1e-3}])
    poses_all = qt_to_transform_matrix(cur_rot, cur_trans)
    opt.zero_grad(), pose_opt.zero_grad()
    losses = []
    for i in range(g["pixel_idx"].shape[0]):
        idx = T(g["pixel_idx"][i])
        r, c = torch.div(idx, W, rounding_mode="floor"), torch.remainder(idx, W)
        d_cam = frame["direction"][r, c].to(dev)
        t_rgb, t_d = frame["rgb"][r, c].to(dev), frame["depth"][r, c][:, None].to(dev)
        which = torch.zeros(idx.shape[0], dtype=torch.int64, device=dev)
        rays_d = torch.sum(d_cam[..., None, None, :] * poses_all[which, None, :3, :3], -1).reshape(-1, 3)
        rays_o = poses_all[which, :3, -1].reshape(-1, 3)
        ret = m.forward(rays_o, rays_d, t_rgb, t_d, noise=T(g["noise"][i]).to(dev))
        loss = (tr["rgb_weight"] * ret["rgb_loss"] + tr["depth_weight"] * ret["depth_loss"]
                + tr["sdf_weight"] * ret["sdf_loss"] + tr["fs_weight"] * ret["fs_loss"])
        loss.backward(retain_graph=True)
        opt.step()
        opt.zero_grad()
        if (i + 1) % 2 == 0:
            pose_opt.step()
            poses_all = qt_to_transform_matrix(cur_rot, cur_trans)
            pose_opt.zero_grad()
        losses.append(float(loss))
    np.testing.assert_allclose(np.array(losses), g["losses"], rtol=2e-3)
    assert_close(cur_trans, g["trans_final"], 1e-4, "optimised translation")
    assert_close(cur_rot, g["rot_final"], 1e-4, "optimised quaternion")
    assert_close(m.decoder.pts_linear[2].weight, g["w1.decoder.pts_linear.2.weight"], 5e-3, "decoder after 6 steps")


def test_hipgraph_replay_matches_eager_iterations(dev):
    """Capturing fwd+bwd+FusedAdam(capturable) into a hipGraph and replaying it must give the parameters eager
    execution gives (same inputs, same number of steps)."""
    from mipsfusion_amd.graph import GraphedSteps
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        inputs = [T(g[k]).to(dev) for k in ("rays_o", "rays_d", "target_rgb", "target_d", "noise")]

        def make():
            m = make_scene(g, cfg, dev).train()
            opt = FusedAdam([{"params": m.decoder.parameters(), "weight_decay": 1e-6, "lr": 0.01},
                             {"params": m.embed_fn.parameters(), "eps": 1e-15, "lr": 0.01}], betas=(0.9, 0.99),
                            capturable=True)

            def step(_k=0):
                ret = m.forward(*inputs[:4], noise=inputs[4])
                path_cpu.total_loss(ret, cfg["training"]).backward()
                opt.step(zero_grad=True)
            return m, step

        m_eager, step_eager = make()
        for _ in range(5):
            step_eager()
        m_graph, step_graph = make()
        graphed = GraphedSteps(step_graph, 1, warmup=2, stream=side)      # 2 warm-up steps ran; capture only records
        for _ in range(3):
            graphed.replay()                                               # 5 steps in total
        torch.cuda.synchronize()
    for (k, a), (_, b) in zip(m_graph.named_parameters(), m_eager.named_parameters()):
        if a.numel():
            assert_grad_close(a, b, 1e-5, "after 5 steps: " + k)


# ------------------------------------------------------------- full-size property checks
def headline_scene(dev, hash_size=19):
    cfg = synth.config_headline()
    cfg["grid"]["hash_size"] = hash_size
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    torch.manual_seed(0)
    m = JointEncoding(cfg, bb, nf).to(dev)
    with torch.no_grad():
        m.embed_fn.params.copy_((torch.randn(m.embed_fn.params.shape) * 0.2).to(dev))
    frame = synth.make_frame(cfg, seed=0)
    H, W = frame["depth"].shape
    random.seed(0)
    idx = torch.tensor(random.sample(range(H * W), 4096))
    batch = [t.to(dev) for t in synth.ray_batch(frame, idx, frame["c2w"])]
    return cfg, m, batch


@pytest.mark.slow
def test_full_size_properties(dev):
    """4096 rays x 64 samples, hash 2^19: properties that do not need the (slow) oracle."""
    cfg, m, (ro, rd, rgb, d) = headline_scene(dev)
    N, S = 4096, 64
    noise = torch.rand(N, S, device=dev)
    m.eval()
    with torch.no_grad():
        out = m.forward(ro, rd, None, d, noise=noise)
    z = out["z_vals"]
    assert z.shape == (N, S) and bool((z[:, 1:] >= z[:, :-1]).all()), "samples sorted along every ray"
    assert bool((out["acc_map"] <= 1.0 + 1e-5).all()) and bool((out["acc_map"] >= 0).all())
    p = out["raw"][..., 5:]
    assert_close(p.sum(-1), torch.ones(N, S), 1e-5, "class probabilities sum to one")
    sdf_from_p = ((p * torch.arange(5.0, device=dev)).sum(-1) / 4 - 0.5) * 2
    assert_close(out["raw"][..., 3], sdf_from_p, 1e-5, "sdf = expectation of the class distribution")
    # determinism of the forward pass
    with torch.no_grad():
        again = m.forward(ro, rd, None, d, noise=noise)
    assert torch.equal(again["raw"], out["raw"]) and torch.equal(again["depth"], out["depth"])
    # linearity of the grid in its parameters (scale table by 2 -> features x2 exactly)
    meta = m.embed_fn.meta
    xn = torch.rand(N * S, 3, device=dev)
    f1 = ops.hashgrid_fwd(xn, m.embed_fn.params.detach(), meta)
    f2 = ops.hashgrid_fwd(xn, (2.0 * m.embed_fn.params.detach()).contiguous(), meta)
    assert torch.equal(f2, 2.0 * f1)
    # partition of unity: a constant table is reproduced, and the scatter conserves the gradient mass per level
    const = torch.full_like(m.embed_fn.params, 0.25)
    assert_close(ops.hashgrid_fwd(xn, const, meta), torch.full((N * S, 32), 0.25), 1e-6, "constant table")
    dy = torch.randn(N * S, 32, device=dev)
    dp = torch.zeros_like(const)
    ops.hashgrid_bwd(xn, const, dy, dp, meta, _lib.FEAT_AOS, None)
    offs = list(meta.offsets[:17])
    for lvl in (0, 5, 9, 15):
        seg = dp[2 * offs[lvl]:2 * offs[lvl + 1]].reshape(-1, 2).double().sum(0)
        ref = dy[:, 2 * lvl:2 * lvl + 2].double().sum(0)
        assert_close(seg, ref, 1e-4, f"gradient mass level {lvl}")


@pytest.mark.slow
def test_full_size_training_step_vs_oracle_subset(dev):
    """Full 4096x64 iteration on the GPU; the oracle re-runs a 192-ray subset with the same parameters and the
    per-ray outputs must agree (losses are global means, so they are compared on the subset run alone)."""
    cfg, m, (ro, rd, rgb, d) = headline_scene(dev, hash_size=16)
    N, S = 4096, 64
    noise = torch.rand(N, S, device=dev)
    m.train()
    ret = m.forward(ro, rd, rgb, d, noise=noise)
    loss = path_cpu.total_loss(ret, cfg["training"])
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(m.embed_fn.params.grad).all()
    sub = slice(0, 192)
    cpu = path_cpu.CpuScene(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"])
    cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    o = cpu.train_forward(ro[sub].cpu(), rd[sub].cpu(), rgb[sub].cpu(), d[sub].cpu(), noise[sub].cpu(), 0.01)
    assert_close(ret["rgb"][sub], o["rgb"], 1e-4, "rgb map")
    assert_close(ret["depth"][sub], o["depth"], 1e-4, "depth map")
    m.zero_grad()
    ret2 = m.forward(ro[sub], rd[sub], rgb[sub], d[sub], noise=noise[sub])
    for k in ("rgb_loss", "sdf_loss", "fs_loss", "depth_loss"):
        assert_close(ret2[k], o[k], 1e-4, k)
    path_cpu.total_loss(ret2, cfg["training"]).backward()
    path_cpu.total_loss(o, cfg["training"]).backward()
    assert_grad_close(m.embed_fn.params.grad, cpu.embed_fn.params.grad, 5e-4, "grid gradient")
    assert_grad_close(m.decoder.sdf_linear[0].weight.grad, cpu.decoder.sdf_linear[0].weight.grad, 5e-4,
                      "decoder gradient")


# ------------------------------------------------------------------------ RandomOptimizer
def _ro_setup(g, dev, particle_size=None):
    import types
    from mipsfusion_amd.RandomOptimizer import RandomOptimizer
    cfg = synth.config_plumbing()
    P = int(g["particle_size"]) if particle_size is None else particle_size
    cfg["tracking"]["RO"] = dict(particle_size=P, n_rows=int(g["n_rows"]), n_cols=int(g["n_cols"]),
                                 initial_scaling_factor=float(g["c1"]), rescaling_factor=float(g["c2"]))
    cfg["tracking"]["ignore_edge_W"] = cfg["tracking"]["ignore_edge_H"] = 2
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=T(g["rays_dir"]))
    ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
    return cfg, ro


@pytest.mark.gpu
def test_random_optimizer_matches_reference_golden(dev):
    """Fused particle rounds (ro_particles -> hashgrid -> decoder -> ro_fitness -> ro_update) vs the poses the
    reference's own RandomOptimizer.optimize produced (tests/golden/ro.npz), round by round."""
    g = load_golden("ro.npz")
    cfg, ro = _ro_setup(g, dev)
    assert ro.decoder_precision == "bf16x6", "the reference-faithful arithmetic is the RandomOptimizer's default"
    assert np.array_equal(ro.row_indices.numpy(), g["rows"]) and np.array_equal(ro.col_indices.numpy(), g["cols"])
    ro.pre_sampled_particle = T(g["pst"]).to(dev).contiguous()
    m = make_scene(g, cfg, dev)
    m.eval()
    depth, init = T(g["depth"]), T(g["init_pose"])
    # first half of a round against the reference's intermediate tensors
    state = torch.zeros(ops.RO_STATE_FLOATS, device=dev)
    state[0:9], state[9:12], state[12:18] = init[:3, :3].reshape(9).to(dev), init[:3, 3].to(dev), float(g["c1"])
    td = depth[ro.row_indices, ro.col_indices].to(dev).contiguous()
    xn, pst7 = ops.ro_particles(ro.pre_sampled_particle, state, ro._dirs[0], td, m._rc(1, 0))
    assert_close(pst7, g["pst7_0"], 1e-7, "7-D particle poses")
    world = (T(g["abs_rot0"]) @ (T(g["rays_dir"])[ro.row_indices, ro.col_indices] * td.cpu()[:, None]).T
             + T(g["abs_trans0"])).transpose(1, 2)
    bound = T(g["bound"]).double()
    xn_ref = ((world.double() - bound[:, 0]) / (bound[:, 1] - bound[:, 0])).float().reshape(-1, 3)
    assert_close(xn, xn_ref, 2e-6, "normalised particle points")
    mm = ops.ro_fitness(m.query_color_sdf(xn).view(pst7.shape[0], -1, 10), td, float(g["trunc"]))
    assert_close(mm, g["mean_masked0"], 1e-4, "mean masked |sdf| per particle")
    # whole optimisation
    for n_iter in range(0, 7):
        pose = ro.optimize(m, depth, init.clone(), None, n_iter=n_iter)
        assert_close(pose, g[f"pose_after_{n_iter}"], 1e-4, f"tracked pose after {n_iter} rounds")


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16", "f32", "f16x3"])
def test_random_optimizer_f16_rounds_track_the_reference_pose(dev, precision):
    """The opt-in plain-f16 matrix-core decoder (BASELINE config 5 "fp16 decoder on CDNA4"; bench.py and the measured
    sequences select it and say so): the pose tracked over 5 rounds (and every other round count of ro.npz) must stay
    within 1e-3 of the pose the reference's own class produced; "f32" (fp32-input MFMA) and "f16x3" (hi/lo f16 operands) are
    held to 1e-4 like the default "bf16x6" above."""
    g = load_golden("ro.npz")
    cfg, ro = _ro_setup(g, dev)
    ro.decoder_precision = precision
    ro.pre_sampled_particle = T(g["pst"]).to(dev).contiguous()
    m = make_scene(g, cfg, dev).eval()
    depth, init = T(g["depth"]), T(g["init_pose"])
    worst = 0.0
    for n_iter in range(0, 7):
        pose = ro.optimize(m, depth, init.clone(), None, n_iter=n_iter)
        worst = max(worst, float((pose.cpu() - T(g[f"pose_after_{n_iter}"])).abs().max()))
    print(f"RandomOptimizer precision {precision}: max |pose - reference pose| over 0..6 rounds = {worst:.2e}")
    assert worst < (1e-3 if precision == "f16" else 1e-4)


@pytest.mark.gpu
def test_random_optimizer_full_size_vs_oracle(dev):
    """2000 particles x (16 x 24) lattice points at the headline grid: three fused rounds vs oracle/ro_cpu.py fed
    with the same network (the oracle queries the GPU model so that only the particle logic is compared)."""
    import types
    from oracle import ro_cpu
    from mipsfusion_amd.RandomOptimizer import RandomOptimizer
    cfg, m, _ = headline_scene(dev, hash_size=16)
    cfg["tracking"]["RO"].update(initial_scaling_factor=0.02, rescaling_factor=0.5)
    cfg["tracking"]["ignore_edge_W"] = cfg["tracking"]["ignore_edge_H"] = 20
    frame = synth.make_frame(cfg, seed=3)
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frame["direction"])
    np.random.seed(5)
    ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
    ro.decoder_precision = m.decoder_precision      # the oracle below queries the GPU model: same arithmetic on both sides
    m.eval()
    init = frame["c2w"].clone()
    init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
    pose, state = ro.optimize(m, frame["depth"], init, None, n_iter=3, return_state=True)

    def net(world):
        with torch.no_grad():
            return m.run_network(world.to(dev)).cpu()
    ref_pose, trace = ro_cpu.optimize(net, ro.pre_sampled_particle.cpu(), ro.row_indices, ro.col_indices,
                                      frame["depth"], frame["direction"], init, 3, 0.02, 0.5,
                                      cfg["training"]["trunc"])
    assert_close(pose, ref_pose, 1e-4, "tracked pose, 3 rounds, 2000 particles")
    assert_close(state[12:18], trace[-1]["search"].reshape(6), 1e-3, "search size")
    assert bool(state[18] > 0) == trace[-1]["success"]


# --------------------------------------------------------------- device-resident keyframe rays
@pytest.mark.gpu
def test_keyframe_ray_sampling_matches_reference_bit_for_bit(dev):
    """DeviceRayDB (HBM database, host index stream, gather kernel) vs the rows the reference's own KeyframeSet
    sampled for the same python RNG seeds (tests/golden/keyframe_rays.npz)."""
    from mipsfusion_amd.keyframe_rays import DeviceRayDB
    g = load_golden("keyframe_rays.npz")
    db = DeviceRayDB(g["db"].shape[0], int(g["num_rays_to_save"]), dev)
    for k in range(g["db"].shape[0]):
        db.store(k, T(g["db"][k]))
    for name in ("sub1", "sub2", "sub5"):
        random.seed(int(g[f"{name}.seed"]))
        rays, kf_ids, kf_indices = db.sample_rays_in_submap(T(g[f"{name}.first"]), T(g[f"{name}.related"]).long(),
                                                            int(g[f"{name}.n"]))
        assert rays.is_cuda and np.array_equal(rays.cpu().numpy(), g[f"{name}.rays"]), name
        assert np.array_equal(kf_ids.numpy(), g[f"{name}.kf_ids"]) and np.array_equal(kf_indices.numpy(),
                                                                                      g[f"{name}.kf_indices"])
    random.seed(7)
    rays, kf_ids, kf_indices = db.sample_rays_in_given_kf(torch.tensor([6, 1, 3]), 30)
    assert np.array_equal(rays.cpu().numpy(), g["given.rays"]) and np.array_equal(kf_ids.numpy(), g["given.kf_ids"])
    assert np.array_equal(kf_indices.numpy(), g["given.kf_indices"])
    random.seed(8)
    rays, kf_ids = db.sample_global_rays(25, g["db"].shape[0])
    assert np.array_equal(rays.cpu().numpy(), g["global.rays"]) and np.array_equal(kf_ids.numpy(), g["global.kf_ids"])
    random.seed(9)
    rays, kf_indices = db.sample_rays_from_given(torch.tensor([5, 0]), 20)
    assert np.array_equal(rays.cpu().numpy(), g["from_given.rays"])
    assert np.array_equal(kf_indices.numpy(), g["from_given.kf_indices"])
    # split form (what mipsfusion.py:315-317 slices and uploads), negative and empty index lists
    idx = torch.tensor([0, 5, -1, 17], device=dev)
    d_cam, rgb, depth = ops.gather_rays(db.rays, idx, split=True)
    flat = T(g["db"]).reshape(-1, 7)
    ref = flat[[0, 5, flat.shape[0] - 1, 17]]
    assert torch.equal(d_cam.cpu(), ref[:, :3]) and torch.equal(rgb.cpu(), ref[:, 3:6])
    assert torch.equal(depth.cpu(), ref[:, 6:7])
    assert ops.gather_rays(db.rays, torch.empty(0, dtype=torch.int64, device=dev)).shape == (0, 7)
    # caller-owned storage: database rows and extra rows (a current frame) in one table, gathered by one index list
    K, R = g["db"].shape[0], int(g["num_rays_to_save"])
    table = torch.zeros(K * R + 4, 7, device=dev)
    db2 = DeviceRayDB(K, R, dev, storage=table)
    db2.store(2, T(g["db"][2]))
    table[K * R:] = 5.0
    rows = ops.gather_rays(table, torch.tensor([2 * R + 1, K * R + 3], device=dev))
    assert torch.equal(rows[0].cpu(), T(g["db"][2][1])) and bool((rows[1] == 5.0).all())
    assert db2.rays.data_ptr() == table.data_ptr()


def test_sdf_only_forward_is_column_3_bit_for_bit(dev):
    """mipsf_decoder_fwd_sdf (the query_sdf column, scene_rep.py:106-107: half of layer 2, no rgb head) vs column 3 of the full
    forward, on the small-batch kernel and on the persistent LDS kernel, ragged sizes; and JointEncoding.query_sdf
    (no-grad path) vs query_color_sdf."""
    torch.manual_seed(11)
    cfg = synth.config_plumbing()
    m = JointEncoding(cfg, T(np.array(cfg["mapping"]["bound"])), T(np.array(cfg["mapping"]["localMLP_max_len"]))).to(dev)
    with torch.no_grad():
        m.embed_fn.params.uniform_(-0.5, 0.5)
        for p in m.decoder.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    packed = ops.decoder_pack(m.decoder.ordered_parameters())
    for M in (1, 33, 1000, 70001, 300000):
        xn = torch.rand(M, 3, device=dev)
        feat = ops.hashgrid_fwd(xn, m.embed_fn.params.detach(), m.embed_fn.meta, ops.FEAT_LEVEL_MAJOR)
        full, _ = ops.decoder_fwd(packed, feat, ops.FEAT_LEVEL_MAJOR, xn, None, M, save=False)
        sdf = ops.decoder_fwd_sdf(packed, feat, ops.FEAT_LEVEL_MAJOR, xn, None, M)
        assert torch.equal(sdf, full[:, 3]), M
    pts = torch.rand(5000, 3, device=dev) * 2 - 1
    with torch.no_grad():
        assert torch.equal(m.query_sdf(pts), m.query_color_sdf(pts)[..., 3:4])


def test_forward_above_16m_samples_matches_chunked_evaluation(dev):
    """Maximum sizes: above 2^24 samples the forward leaves the single 4 GB buffer resource for the grid features and
    takes its 64-bit addressing branch; a batch of 2^24 + 77 samples must equal the same points evaluated in chunks
    (hash grid + full decoder forward + SDF-only forward), bit for bit."""
    torch.manual_seed(12)
    cfg = synth.config_plumbing()
    m = JointEncoding(cfg, T(np.array(cfg["mapping"]["bound"])), T(np.array(cfg["mapping"]["localMLP_max_len"]))).to(dev)
    with torch.no_grad():
        m.embed_fn.params.uniform_(-0.5, 0.5)
    packed = ops.decoder_pack(m.decoder.ordered_parameters())
    M = (1 << 24) + 77
    xn = torch.rand(M, 3, device=dev)
    feat = ops.hashgrid_fwd(xn, m.embed_fn.params.detach(), m.embed_fn.meta, ops.FEAT_LEVEL_MAJOR)
    sdf = ops.decoder_fwd_sdf(packed, feat, ops.FEAT_LEVEL_MAJOR, xn, None, M)
    full, _ = ops.decoder_fwd(packed, feat, ops.FEAT_LEVEL_MAJOR, xn, None, M, save=False)
    assert torch.equal(sdf, full[:, 3])
    del feat
    for lo, hi in ((0, 100000), (M - 100077, M)):
        xc = xn[lo:hi].contiguous()
        fc = ops.hashgrid_fwd(xc, m.embed_fn.params.detach(), m.embed_fn.meta, ops.FEAT_LEVEL_MAJOR)
        oc, _ = ops.decoder_fwd(packed, fc, ops.FEAT_LEVEL_MAJOR, xc, None, hi - lo, save=False)
        assert torch.equal(oc, full[lo:hi]), (lo, hi)


@pytest.mark.gpu
def test_reference_checkpoint_reproduces_reference_outputs(dev):
    """the checkpoint file written by the reference (tests/golden/ref_model_0.pth) loaded into the GPU
    model must give the outputs the reference model gave for it (tests/golden/checkpoint_probe.npz)."""
    from mipsfusion_amd import checkpoint
    g = load_golden("checkpoint_probe.npz")
    cfg = synth.config_plumbing()
    m = JointEncoding(cfg, T(g["bound"]), T(g["half_len"])).to(dev)
    checkpoint.load_state_dict(m, os.path.join(GOLDEN, "ref_model_0.pth"))
    assert m.embed_fn.params.is_cuda
    assert_close(m.query_color_sdf(T(g["pts"]).to(dev)), g["out"], 1e-4, "query_color_sdf of the loaded checkpoint")


# ------------------------------------------------------------------ inference consumers
@pytest.mark.gpu
def test_full_image_render_and_grid_queries(dev):
    """Logger.render_full_img-style chunked rendering of a whole (32x32) frame and Mesher-style batched queries:
    equal to the oracle, independent of the chunk size, and equal to the stitched shares of a 3-way ray split."""
    from mipsfusion_amd import inference
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    m = make_scene(g, cfg, dev)
    m.eval()
    cpu = path_cpu.CpuScene(cfg, g["bound"], g["half_len"])
    cpu.load_state_dict({k[2:]: T(g[k]) for k in g.files if k.startswith("w.")})
    frame = synth.make_frame(cfg, seed=2)
    H, W = frame["depth"].shape
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    torch.manual_seed(3)
    noise = torch.rand(H * W, S)
    rgb, depth = inference.render_full_img(m, frame["direction"], frame["c2w"], frame["depth"], H, W,
                                           ray_batch_size=300, noise=noise.to(dev))
    assert rgb.shape == (H, W, 3) and depth.shape == (H, W)
    rays_d, rays_o = inference.rays_camera_to_world(frame["direction"].reshape(-1, 3), frame["c2w"])
    with torch.no_grad():
        ref = cpu.render_rays(rays_o, rays_d, frame["depth"].reshape(-1, 1), noise)
    assert_close(rgb.reshape(-1, 3), ref["rgb"], 1e-4, "full-image rgb")
    assert_close(depth.reshape(-1), ref["depth"], 1e-4, "full-image depth")
    rgb2, depth2 = inference.render_full_img(m, frame["direction"], frame["c2w"], frame["depth"], H, W,
                                             ray_batch_size=10000, noise=noise.to(dev))
    assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2), "chunking must not change a single pixel"
    # the ray-data-parallel split renders exactly the same pixels (shares stitched by hand: one process here)
    parts = []
    rd_g, ro_g = inference.rays_camera_to_world(frame["direction"].reshape(-1, 3).to(dev), frame["c2w"].to(dev))
    for r in range(3):
        b, e = inference.share_of(H * W, r, 3)
        with torch.no_grad():
            out = m.render_rays(ro_g[b:e].contiguous(), rd_g[b:e].contiguous(),
                                frame["depth"].reshape(-1, 1)[b:e].to(dev), noise=noise[b:e].to(dev))
        parts.append(out["rgb"])
    assert torch.equal(torch.cat(parts, 0), rgb.reshape(-1, 3))
    assert inference.share_of(10, 0, 3) == (0, 4) and inference.share_of(10, 2, 3) == (7, 10)
    # Mesher-style dense grid queries on pre-normalised coordinates
    lin = torch.linspace(0.02, 0.98, 12)
    grid = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)
    with torch.no_grad():
        refq = cpu.query_normalised(grid)
    for name, sl in (("query_sdf", slice(3, 4)), ("query_sdf_entropy_prob", slice(3, 10)), ("query_color_sdf", slice(0, 10))):
        out = inference.query_in_batches(getattr(m, name), grid.to(dev), batch_size=500)
        assert_close(out, refq[:, sl], 1e-4, name)
    assert_close(inference.query_in_batches(m.query_color, grid.to(dev), batch_size=700), torch.sigmoid(refq[:, :3]),
                 1e-4, "query_color")


def test_autograd_grad_and_partial_backward_with_trainable_params(dev):
    """by default parameter gradients flow THROUGH autograd, so torch.autograd.grad(sdf, pts) with
    trainable parameters leaves every .grad untouched, autograd.grad(loss, params) returns the gradients, tensor
    hooks fire; the opt-in in-place accumulation gives the same .grad values for a plain backward()."""
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    m = make_scene(g, cfg, dev).train()
    assert m.accumulate_param_grads_in_place is False
    pts = torch.rand(200, 3, device=dev, requires_grad=True)
    sdf = m.query_sdf(pts)
    (normals,) = torch.autograd.grad(sdf.sum(), pts)                 # eikonal / normal style call
    assert normals.shape == pts.shape and torch.isfinite(normals).all() and normals.abs().max() > 0
    assert all(p.grad is None for p in m.parameters()), "autograd.grad wrt the points must not touch parameter .grad"
    inputs = [T(g[k]).to(dev) for k in ("rays_o", "rays_d", "target_rgb", "target_d", "noise")]
    ret = m.forward(*inputs[:4], noise=inputs[4])
    loss = path_cpu.total_loss(ret, cfg["training"])
    params = [p for p in m.parameters() if p.numel()]
    grads = torch.autograd.grad(loss, params, retain_graph=True)
    assert all(gr is not None and torch.isfinite(gr).all() for gr in grads)
    assert all(p.grad is None for p in params)
    hits = []
    h = m.embed_fn.params.register_hook(lambda gr: hits.append(float(gr.abs().sum())))
    loss.backward()
    h.remove()
    assert len(hits) == 1 and hits[0] > 0, "tensor hook on the grid parameters fires on backward()"
    for p, gr in zip(params, grads):
        assert_grad_close(p.grad, gr, 1e-6, "backward() vs autograd.grad")
    # opt-in in-place accumulation: same values
    m2 = make_scene(g, cfg, dev).train()
    m2.accumulate_param_grads_in_place = True
    ret2 = m2.forward(*inputs[:4], noise=inputs[4])
    path_cpu.total_loss(ret2, cfg["training"]).backward()
    for p, q in zip(params, [q for q in m2.parameters() if q.numel()]):
        assert_grad_close(q.grad, p.grad, 1e-6, "in-place accumulation vs autograd path")


def test_route_ahead_on_second_stream_gives_identical_gradients(dev, monkeypatch):
    """JointEncoding.route_ahead (opt-in): the routing half of the hash grid's backward runs on a second stream next to
    the forward (mipsf_hashgrid_route + mipsf_hashgrid_bwd with MIPSF_HG_ROUTED instead of mipsf_hashgrid_bwd).  Same kernels on the
    same data: every gradient must equal the single-stream path's (to the run-to-run noise of the scatter's atomics),
    eagerly and inside a captured graph."""
    from mipsfusion_amd.model import scene_rep
    monkeypatch.setattr(scene_rep, "_ROUTE_AHEAD_MIN_M", 0)
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    inputs = [T(g[k]).to(dev) for k in ("rays_o", "rays_d", "target_rgb", "target_d", "noise")]

    def grads(route_ahead):
        m = make_scene(g, cfg, dev).train()
        m.route_ahead = route_ahead
        ret = m.forward(*inputs[:4], noise=inputs[4])
        path_cpu.total_loss(ret, cfg["training"]).backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in m.parameters() if p.numel()]

    ref, ahead = grads(False), grads(True)
    assert float(ref[0].abs().max()) > 0
    for a, b in zip(ahead, ref):       # (fp64 LDS atomics: the order of additions may differ in the last fp32 bit run to run)
        assert_grad_close(a, b, 1e-6, "route-ahead vs single-stream")
    # captured: the second stream forks from and re-joins the capturing stream inside the forward
    m = make_scene(g, cfg, dev).train()
    m.route_ahead = True
    m.accumulate_param_grads_in_place = True
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        def step():
            ret = m.forward(*inputs[:4], noise=inputs[4])
            path_cpu.total_loss(ret, cfg["training"]).backward()
        step()                                                     # allocator warm-up
        for prm in m.parameters():
            prm.grad = None
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            step()
        for prm in m.parameters():
            if prm.grad is not None:
                prm.grad.zero_()
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip([p.grad for p in m.parameters() if p.numel()], ref):
        assert_grad_close(a, b, 1e-6, "captured route-ahead vs single-stream")


def test_objective_formed_inside_the_loss_kernel_equals_get_loss_from_ret(dev):
    """JointEncoding.forward also returns the weighted objective formed by the loss kernel ("_loss_total", the model's own
    config weights); get_loss_from_ret hands it out when asked for exactly those weights with all four terms on.  Value
    and gradients must equal the explicit weighted sum (other weights / a term switched off take the dot-product path)."""
    from mipsfusion_amd.helper_functions.utils import get_loss_from_ret
    g = load_golden("scene_cfg1.npz")
    cfg = cfg_for("scene_cfg1.npz")
    inputs = [T(g[k]).to(dev) for k in ("rays_o", "rays_d", "target_rgb", "target_d", "noise")]
    tr = cfg["training"]

    def run(how):
        m = make_scene(g, cfg, dev).train()
        ret = m.forward(*inputs[:4], noise=inputs[4])
        assert "_loss_total" in ret
        if how == "kernel":
            loss = get_loss_from_ret(ret, tr)
            assert loss is ret["_loss_total"]
        elif how == "explicit":
            loss = tr["rgb_weight"] * ret["rgb_loss"] + tr["depth_weight"] * ret["depth_loss"] + \
                tr["sdf_weight"] * ret["sdf_loss"] + tr["fs_weight"] * ret["fs_loss"]
        else:                                   # other weights: must NOT take the kernel's total
            other = dict(tr, rgb_weight=tr["rgb_weight"] * 2.0)
            loss = get_loss_from_ret(ret, other)
            assert loss is not ret["_loss_total"]
            return float(loss), None
        loss.backward()
        return float(loss), [p.grad.clone() for p in m.parameters() if p.numel()]

    lk, gk = run("kernel")
    le, ge = run("explicit")
    lo, _ = run("other")
    assert abs(lk - le) <= 2e-6 * abs(le), (lk, le)
    assert abs(lo - le) > 1e-4 * abs(le)
    for a, b in zip(gk, ge):
        assert_grad_close(a, b, 1e-6, "objective from the kernel vs explicit weighted sum")


def test_out_of_range_ray_index_is_loud(dev):
    """a bad keyframe id must not become a silent out-of-bounds read -- host index tensors raise
    IndexError like the reference's torch indexing, device-resident indices yield NaN rays."""
    from mipsfusion_amd.keyframe_rays import DeviceRayDB
    db = DeviceRayDB(2, 10, dev)
    db.store(0, torch.ones(10, 7))
    with pytest.raises(IndexError):
        db._gather(torch.tensor([0, 25]))
    out = ops.gather_rays(db.rays, torch.tensor([3, 20, -1, -21], device=dev))
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).all() and torch.isfinite(out[2]).all() \
        and torch.isnan(out[3]).all()
    rot = torch.tensor([[1., 0, 0, 0]], device=dev)
    ro_, rd_ = ops.pose_rays(rot, torch.zeros(1, 3, device=dev), None, torch.tensor([0, 1, -1, -2], device=dev),
                             torch.ones(4, 3, device=dev))
    assert torch.isfinite(rd_[0]).all() and torch.isnan(rd_[1]).all() and torch.isfinite(rd_[2]).all() \
        and torch.isnan(rd_[3]).all()


def test_fused_adam_capturable_state_dict_roundtrip(dev):
    """with capturable=True the live step counter is a device tensor; state_dict() must carry it and
    load_state_dict() must restore it (bias correction after a restore equals an uninterrupted run)."""
    torch.manual_seed(0)
    p0 = torch.randn(5000, device=dev)
    grads = [torch.randn(5000, device=dev) for _ in range(6)]

    def run(n_first, reload):
        p = torch.nn.Parameter(p0.clone())
        opt = FusedAdam([p], lr=1e-2, betas=(0.9, 0.99), capturable=True)
        for k in range(n_first):
            p.grad = grads[k].clone()
            opt.step()
        if reload:
            sd = opt.state_dict()
            assert sd["state"][0]["step"] == n_first
            opt = FusedAdam([p], lr=1e-2, betas=(0.9, 0.99), capturable=True)
            opt.load_state_dict(sd)
        for k in range(n_first, 6):
            p.grad = grads[k].clone()
            opt.step()
        return p.detach().clone()
    assert torch.equal(run(3, False), run(3, True))


# ------------------------------------------------------------------ f16-MFMA decoder forward (csrc/decoder16.hip)
@pytest.mark.parametrize("M", [1, 33, 1000, 70000])
@pytest.mark.parametrize("layout", ["aos", "level_major"])
def test_decoder_f16x3_forward_matches_fp32_kernel_and_oracle(dev, M, layout):
    """precision "f16x3" (hi/lo split operands on v_mfma_f32_32x32x16_f16) vs the fp32-MFMA kernel and the oracle:
    outputs to fp32 round-off class, the saved-activation record element for element (same layout: the fp32 backward
    kernels consume it unchanged).  Small-batch and persistent (M = 70000) variants, both feature layouts."""
    torch.manual_seed(M)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    with torch.no_grad():
        dec.sdf_linear[2].weight.mul_(4.0)
    ws = dec.ordered_parameters()
    packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
    x = torch.rand(M, 3, device=dev)
    feat_aos = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
    lay = _lib.FEAT_AOS if layout == "aos" else _lib.FEAT_LEVEL_MAJOR
    feat = feat_aos if layout == "aos" else feat_aos.view(M, 16, 2).permute(1, 0, 2).contiguous()
    o32, s32 = ops.decoder_fwd(packed, feat, lay, x, None, M, save=True)
    o16, s16 = ops.decoder_fwd(packed, feat, lay, x, None, M, save=True, precision="f16x3", packed16=packed16)
    assert_close(o16, o32, 2e-6, "f16x3 vs fp32 kernel, [M,10] output")
    n_act = ((M + 127) // 128) * 4 * 192 * 64
    n_tiles = (M + 31) // 32
    a32 = s32[:n_act].view(-1, 192 * 64)[:n_tiles]
    a16 = s16[:n_act].view(-1, 192 * 64)[:n_tiles]
    assert_close(a16, a32, 2e-6, "saved activations (H1, H2, H3 accumulator images)")
    w = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    pe = tcnn_cpu.frequency_forward(x.cpu(), 8)
    ref = path_cpu.decoder_forward(w, feat_aos.cpu(), pe, x.cpu())
    assert_close(o16, ref, 3e-6, "f16x3 vs oracle")
    sdf16 = ops.decoder_fwd_sdf(packed, feat, lay, x, None, M, precision="f16x3", packed16=packed16)
    assert torch.equal(sdf16, o16[:, 3]), "f16x3 SDF-only branch = column 3 of the full f16x3 forward, bit for bit"
    # plain f16 operands: forward-only, stated tolerance 2e-3 of the output range
    p16, none = ops.decoder_fwd(packed, feat, lay, x, None, M, save=False, precision="f16", packed16=packed16)
    assert none is None
    assert_close(p16, ref, 2e-3, "plain f16 vs oracle")
    sdfp = ops.decoder_fwd_sdf(packed, feat, lay, x, None, M, precision="f16", packed16=packed16)
    assert_close(sdfp, ref[:, 3], 2e-3, "plain f16 SDF-only vs oracle")
    assert float((sdfp - p16[:, 3]).abs().max()) <= 1e-6, "SDF-only branch = column 3 of the full plain-f16 forward"


@pytest.mark.parametrize("M", [64, 4096 + 17, 70000])
@pytest.mark.parametrize("layout", ["aos", "level_major"])
def test_decoder_backward_short_cuts_zero_gradient_tiles_exactly(dev, M, layout):
    """ops.decoder_bwd with the zero-tile flags (mipsf_decoder_bwd_chain16 + mipsf_decoder_wgrad16) against the
    same call with every tile processed: the incoming gradient is zero on the tail of every 64-sample ray (as behind the
    truncation band, scene_rep.py:58-78), on some whole rays and on scattered single samples.  d(features) and d(x) must
    be EQUAL (the skipped tiles are zeros either way, the others go through the same arithmetic); the parameter gradients
    agree to fp32 class (the tiles are dealt to the weight-gradient workgroups in a different order); a frozen decoder
    (no weight gradients) takes the short cut too; a batch without any gradient gives zeros."""
    torch.manual_seed(7 + M)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    ws = dec.ordered_parameters()
    packed16 = ops.decoder_pack16(ws)
    x = torch.rand(M, 3, device=dev)
    feat_aos = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
    lay = _lib.FEAT_AOS if layout == "aos" else _lib.FEAT_LEVEL_MAJOR
    feat = feat_aos if layout == "aos" else feat_aos.view(M, 16, 2).permute(1, 0, 2).contiguous()
    out, saved = ops.decoder_fwd(None, feat, lay, x, None, M, save="lean", precision="f16x3", packed16=packed16)
    dout = torch.randn(M, 10, device=dev) * torch.exp(torch.empty(M, 1, device=dev).uniform_(-16.0, -6.0))
    pos = torch.arange(M, device=dev) % 64
    tail = torch.randint(10, 60, ((M + 63) // 64,), device=dev).repeat_interleave(64)[:M]
    dout[pos >= tail] = 0.0                                      # ray tails
    dout[(torch.arange(M, device=dev) // 64) % 5 == 3] = 0.0     # whole rays
    dout[::11] = 0.0                                             # single samples inside live tiles

    def run(skip, grads):
        keep = ops.SKIP_ZERO_TILES
        ops.SKIP_ZERO_TILES = skip
        try:
            return ops.decoder_bwd(None, feat, lay, x, None, out, dout, saved, grads, M, precision="f16x3",
                                   packed16=packed16)
        finally:
            ops.SKIP_ZERO_TILES = keep

    g_all, g_skip = [torch.zeros_like(w) for w in ws], [torch.zeros_like(w) for w in ws]
    df_all, dx_all, _ = run(False, g_all)
    df_skip, dx_skip, _ = run(True, g_skip)
    assert torch.equal(df_skip, df_all) and torch.equal(dx_skip, dx_all)
    n_zero_tiles = int((dout.view(-1, 10)[:(M // 32) * 32].view(-1, 32 * 10) == 0).all(1).sum())
    assert M < 100 or n_zero_tiles > M // 32 // 5, "the test is meant to exercise the short cut"
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, g_skip, g_all):
        assert_close(a, b, 1e-6, "zero-tile short cut, grad " + k)
    df_frozen, dx_frozen, _ = run(True, None)
    assert torch.equal(df_frozen, df_all) and torch.equal(dx_frozen, dx_all)
    # the record of a frozen decoder's forward: the ReLU masks only (all the chain reads of it) -- same outputs, same chain
    out_m, saved_m = ops.decoder_fwd(None, feat, lay, x, None, M, save="masks", precision="f16x3", packed16=packed16)
    assert torch.equal(out_m, out)
    df_m, dx_m, _ = ops.decoder_bwd(None, feat, lay, x, None, out_m, dout, saved_m, None, M, precision="f16x3", packed16=packed16)
    assert torch.equal(df_m, df_all) and torch.equal(dx_m, dx_all)
    with pytest.raises(RuntimeError, match="masks only"):
        ops.decoder_bwd(None, feat, lay, x, None, out_m, dout, saved_m, [torch.zeros_like(w) for w in ws], M,
                        precision="f16x3", packed16=packed16)
    dout.zero_()
    g0 = [torch.zeros_like(w) for w in ws]
    df0, dx0, _ = run(True, g0)
    assert not df0.any() and not dx0.any() and all(not g.any() for g in g0)


@pytest.mark.parametrize("M", [1, 33, 1000, 70000])
@pytest.mark.parametrize("layout", ["aos", "level_major"])
def test_decoder_f16x3_backward_chain_matches_fp32_kernel(dev, M, layout):
    """mipsf_decoder_bwd_chain16 (f16 matrix cores, hi/lo split) vs the fp32-MFMA chain on the same saved record:
    d(features), d(x), the `dact` record (dG3, dH2, dG1 accumulator images + d logits / d rgb) and -- through the
    unchanged weight-gradient kernel -- all ten parameter gradients."""
    torch.manual_seed(100 + M)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    with torch.no_grad():
        dec.sdf_linear[2].weight.mul_(4.0)
    ws = dec.ordered_parameters()
    packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
    x = torch.rand(M, 3, device=dev)
    feat_aos = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
    lay = _lib.FEAT_AOS if layout == "aos" else _lib.FEAT_LEVEL_MAJOR
    feat = feat_aos if layout == "aos" else feat_aos.view(M, 16, 2).permute(1, 0, 2).contiguous()
    out, saved = ops.decoder_fwd(packed, feat, lay, x, None, M, save=True)
    # loss gradients as they really are: a mean over N*S samples puts them at 1e-7 .. 1e-3 (far below f16's normal
    # range: the kernel rescales per sample), a few samples with none at all
    dout = torch.randn(M, 10, device=dev) * torch.exp(torch.empty(M, 1, device=dev).uniform_(-16.0, -6.0))
    dout[::7] = 0.0
    g32 = [torch.zeros_like(w) for w in ws]
    g16 = [torch.zeros_like(w) for w in ws]
    df32, dx32, _ = ops.decoder_bwd(packed, feat, lay, x, None, out, dout, saved, g32, M)
    df16, dx16, _ = ops.decoder_bwd(None, feat, lay, x, None, out, dout, saved, g16, M, precision="f16x3",
                                    packed16=packed16)
    assert_close(df16, df32, 3e-6, "d grid features")
    assert_close(dx16, dx32, 3e-6, "d x")
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, g16, g32):
        assert_close(a, b, 3e-6, "grad " + k)
    # every weight-gradient kernel reads the same records: the default above is the streaming f16 hi/lo kernel
    # (wgrad16.hip); the fp32 LDS kernel and the streaming kernel on three bf16 planes must agree with it to fp32 class,
    # and the streaming kernel must equally work behind the fp32 chain
    for wp in ("f32", "stream_bf16x6"):
        gw = [torch.zeros_like(w) for w in ws]
        ops.decoder_bwd(None, feat, lay, x, None, out, dout, saved, gw, M, precision="f16x3", packed16=packed16,
                        wgrad_precision=wp)
        for k, a, b in zip(ops.DECODER_PARAM_ORDER, gw, g32):
            assert_close(a, b, 3e-6, wp + " grad " + k)
    # the lean record: the f16x3 forward leaves H1 out, the streaming kernel recomputes it from x with the forward's own
    # layer-1 operand images (the final reduction adds its 8 slices atomically: equal up to the last bit only)
    out16, saved16 = ops.decoder_fwd(None, feat, lay, x, None, M, save=True, precision="f16x3", packed16=packed16)
    outl, savedl = ops.decoder_fwd(None, feat, lay, x, None, M, save="lean", precision="f16x3", packed16=packed16)
    assert torch.equal(outl, out16)
    g_full, g_rc, g_lean = ([torch.zeros_like(w) for w in ws] for _ in range(3))
    ops.decoder_bwd(None, feat, lay, x, None, out16, dout, saved16, g_full, M, precision="f16x3", packed16=packed16,
                    wgrad_precision="stream_f16x3")
    ops.decoder_bwd(None, feat, lay, x, None, out16, dout, saved16, g_rc, M, precision="f16x3", packed16=packed16,
                    wgrad_precision="stream_f16x3", recompute_h1=True)
    ops.decoder_bwd(None, feat, lay, x, None, outl, dout, savedl, g_lean, M, precision="f16x3", packed16=packed16,
                    wgrad_precision="stream_f16x3", recompute_h1=True)
    with pytest.raises(RuntimeError, match="lean"):            # a kernel that reads H1 must refuse the lean record
        ops.decoder_bwd(None, feat, lay, x, None, outl, dout, savedl, [torch.zeros_like(w) for w in ws], M,
                        precision="f16x3", packed16=packed16, wgrad_precision="f32")
    for k, a, b, c in zip(ops.DECODER_PARAM_ORDER, g_full, g_rc, g_lean):
        assert_close(b, a, 1e-6, "recomputed H1 vs stored H1, grad " + k)
        assert_close(c, b, 1e-6, "lean record vs full record (H1 is not read either way), grad " + k)
    # the lean GRADIENT record (the default behind recompute_h1: the chain leaves out dG3 and the rgb_emb half of dH2, the
    # exchange form of the weight-gradient kernel recomputes them with the chain's own operations) against the full one
    # through the same kernel: the same values enter the same products (the final reduction adds its slices atomically)
    g_fd = [torch.zeros_like(w) for w in ws]
    keep = ops.LEAN_DACT
    ops.LEAN_DACT = False
    try:
        ops.decoder_bwd(None, feat, lay, x, None, outl, dout, savedl, g_fd, M, precision="f16x3", packed16=packed16,
                        wgrad_precision="stream_f16x3", recompute_h1=True)
    finally:
        ops.LEAN_DACT = keep
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, g_lean, g_fd):
        assert_close(a, b, 1e-6, "lean gradient record vs full gradient record, grad " + k)
    # (g_full is not compared with g32 here: this block differentiates the f16x3 forward, whose ReLU masks differ from
    # the fp32 forward's in a few of 9 M decisions)
    gw = [torch.zeros_like(w) for w in ws]
    ops.decoder_bwd(packed, feat, lay, x, None, out, dout, saved, gw, M, wgrad_precision="stream_f16x3")
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, gw, g32):
        assert_close(a, b, 3e-6, "stream_f16x3 behind the fp32 chain, grad " + k)
    # opt-in: the three large weight-gradient products on the bf16 matrix cores with hi/lo split operands (16-17 bits
    # per operand, fp32 accumulate): per-term error 2^-16, zero-mean, averaged over the batch -> ~5e-6 of the maximum
    gbf = [torch.zeros_like(w) for w in ws]
    ops.decoder_bwd(None, feat, lay, x, None, out, dout, saved, gbf, M, precision="f16x3", packed16=packed16,
                    wgrad_precision="bf16x3")
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, gbf, g32):
        assert_close(a, b, 2e-5, "bf16x3 grad " + k)


# ------------------------------------------------------------------ bf16x6: fp32 operands as three bf16 pieces, six products
@pytest.mark.parametrize("M", [1, 33, 1000, 70000])
@pytest.mark.parametrize("layout", ["aos", "level_major"])
def test_decoder_bf16x6_matches_fp32_kernel_and_oracle(dev, M, layout):
    """precision "bf16x6" (csrc/decoder16.hip, NP = 3: every operand carried exactly as p0 + p1 + p2 in bf16, a product =
    p0 p0 + p0 p1 + p1 p0 + p1 p1 + p0 p2 + p2 p0 on v_mfma_f32_32x32x16_bf16) against the fp32-input MFMA kernels and the
    oracle -- the arithmetic of model/decoder.py:32-50's fp32 nn.Linear layers.  Forward: outputs, the saved record element
    for element, the SDF-only branch bit for bit; every record form (full, lean, masks) gives the same outputs.  Backward:
    the chain on the SAME record as the fp32 chain (d features, d x), the streaming weight-gradient kernel on three bf16
    planes behind it, the zero-tile short cut, a frozen decoder.  Small-batch and persistent (M = 70000) kernels."""
    torch.manual_seed(300 + M)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    with torch.no_grad():
        dec.sdf_linear[2].weight.mul_(4.0)
    ws = dec.ordered_parameters()
    packed, pk = ops.decoder_pack(ws), ops.decoder_pack16(ws, precision="bf16x6")
    x = torch.rand(M, 3, device=dev)
    feat_aos = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
    lay = _lib.FEAT_AOS if layout == "aos" else _lib.FEAT_LEVEL_MAJOR
    feat = feat_aos if layout == "aos" else feat_aos.view(M, 16, 2).permute(1, 0, 2).contiguous()
    o32, s32 = ops.decoder_fwd(packed, feat, lay, x, None, M, save=True)
    ob, sb = ops.decoder_fwd(None, feat, lay, x, None, M, save=True, precision="bf16x6", packed16=pk)
    assert_close(ob, o32, 1e-6, "bf16x6 vs fp32 kernel, [M,10] output")
    n_act = ((M + 127) // 128) * 4 * 192 * 64
    n_tiles = (M + 31) // 32
    assert_close(sb[:n_act].view(-1, 192 * 64)[:n_tiles], s32[:n_act].view(-1, 192 * 64)[:n_tiles], 1e-6, "saved activations")
    w = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    ref = path_cpu.decoder_forward(w, feat_aos.cpu(), tcnn_cpu.frequency_forward(x.cpu(), 8), x.cpu())
    assert_close(ob, ref, 2e-6, "bf16x6 vs oracle")
    assert torch.equal(ops.decoder_fwd_sdf(None, feat, lay, x, None, M, precision="bf16x6", packed16=pk), ob[:, 3])
    for save in (False, "lean", "masks"):
        o2, _ = ops.decoder_fwd(None, feat, lay, x, None, M, save=save, precision="bf16x6", packed16=pk)
        assert torch.equal(o2, ob), f"save={save!r}: same outputs"
    with pytest.raises(RuntimeError, match="packed for"):          # an f16 buffer must not reach the bf16 kernels
        ops.decoder_fwd(None, feat, lay, x, None, M, save=False, precision="bf16x6", packed16=ops.decoder_pack16(ws))
    # ---- backward on the fp32 forward's record: loss gradients of 1e-7 .. 1e-3, some samples without any
    dout = torch.randn(M, 10, device=dev) * torch.exp(torch.empty(M, 1, device=dev).uniform_(-16.0, -6.0))
    dout[::7] = 0.0
    g32, gb = ([torch.zeros_like(w_) for w_ in ws] for _ in range(2))
    df32, dx32, _ = ops.decoder_bwd(packed, feat, lay, x, None, o32, dout, s32, g32, M)
    dfb, dxb, _ = ops.decoder_bwd(None, feat, lay, x, None, o32, dout, s32, gb, M, precision="bf16x6", packed16=pk)
    assert_close(dfb, df32, 1e-6, "d grid features")
    assert_close(dxb, dx32, 1e-6, "d x")
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, gb, g32):
        assert_close(a, b, 2e-6, "grad " + k)
    # the fp32 LDS weight-gradient kernel behind the bf16x6 chain (same `dact` record)
    gl = [torch.zeros_like(w_) for w_ in ws]
    ops.decoder_bwd(None, feat, lay, x, None, o32, dout, s32, gl, M, precision="bf16x6", packed16=pk, wgrad_precision="f32")
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, gl, g32):
        assert_close(a, b, 2e-6, "fp32 wgrad behind the bf16x6 chain, grad " + k)
    # ---- zero tiles: ray tails without a gradient are short-cut, the result must equal the dense evaluation
    dz = dout.clone()
    dz.view(-1)[(torch.arange(M * 10, device=dev) // 10 % 64) >= 31] = 0.0
    keep = ops.SKIP_ZERO_TILES
    res = {}
    for skip in (False, True):
        ops.SKIP_ZERO_TILES = skip
        try:
            g = [torch.zeros_like(w_) for w_ in ws]
            df, dx_, _ = ops.decoder_bwd(None, feat, lay, x, None, ob, dz, sb, g, M, precision="bf16x6", packed16=pk)
            res[skip] = (df, dx_, g)
        finally:
            ops.SKIP_ZERO_TILES = keep
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1]), "short cut: same d feat / d x"
    # (the live-tile lists are filled through atomics: which workgroup sums which tiles -- the order of the fp32 additions of
    # the weight gradients -- changes from run to run and with the short cut; 70 000 samples: 5e-7 typically, 1.03e-6 seen in 1
    # run of 16 on `sdf_linear.2.weight`; the gate is the one of this test's other gradient comparisons)
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, res[True][2], res[False][2]):
        assert_close(a, b, 2e-6, "short cut, grad " + k)
    # ---- the lean records (the default of JointEncoding): the forward leaves H1 out, the chain dG3 and the rgb_emb half of dH2;
    #      the exchange form of the weight-gradient kernel (csrc/wgrad16.hip, three bf16 planes) recomputes all three with the
    #      forward's / the chain's own products in their own order.  Full record through the plain streaming kernel, full record
    #      with H1 recomputed, lean activation record, lean activation + full gradient record: the same gradients
    ol, sl = ops.decoder_fwd(None, feat, lay, x, None, M, save="lean", precision="bf16x6", packed16=pk)
    assert torch.equal(ol, ob)
    kw = dict(precision="bf16x6", packed16=pk, wgrad_precision="stream_bf16x6")
    g_full, g_rc, g_lean, g_fd = ([torch.zeros_like(w_) for w_ in ws] for _ in range(4))
    ops.decoder_bwd(None, feat, lay, x, None, ob, dout, sb, g_full, M, **kw)
    ops.decoder_bwd(None, feat, lay, x, None, ob, dout, sb, g_rc, M, recompute_h1=True, **kw)
    df_l, dx_l, _ = ops.decoder_bwd(None, feat, lay, x, None, ol, dout, sl, g_lean, M, recompute_h1=True, **kw)
    keep_ld = ops.LEAN_DACT
    ops.LEAN_DACT = False
    try:
        ops.decoder_bwd(None, feat, lay, x, None, ol, dout, sl, g_fd, M, recompute_h1=True, **kw)
    finally:
        ops.LEAN_DACT = keep_ld
    with pytest.raises(RuntimeError, match="lean"):            # a kernel that reads H1 must refuse the lean record
        ops.decoder_bwd(None, feat, lay, x, None, ol, dout, sl, [torch.zeros_like(w_) for w_ in ws], M, precision="bf16x6",
                        packed16=pk, wgrad_precision="f32")
    for k, a, b, c, d in zip(ops.DECODER_PARAM_ORDER, g_full, g_rc, g_lean, g_fd):
        assert_close(b, a, 1e-6, "recomputed H1 vs stored H1, grad " + k)
        assert_close(c, b, 1e-6, "lean records vs full records, grad " + k)
        assert_close(d, c, 1e-6, "full gradient record vs lean gradient record, grad " + k)
    # ---- a frozen decoder: masks-only record, no gradient record, no weight gradients
    om, sm = ops.decoder_fwd(None, feat, lay, x, None, M, save="masks", precision="bf16x6", packed16=pk)
    dfm, dxm, _ = ops.decoder_bwd(None, feat, lay, x, None, om, dz, sm, None, M, precision="bf16x6", packed16=pk)
    assert torch.equal(dfm, res[True][0]) and torch.equal(dxm, res[True][1]), "frozen decoder: same d feat / d x"


@pytest.mark.parametrize("precision", ["bf16x6", "f16x3"])
def test_decoder_backward_with_subnormal_incoming_gradients_stays_finite(dev, precision):
    """The chain scales every sample's incoming gradient by the power of two that brings its largest component to [0.5, 1).
    For a sample whose WHOLE incoming gradient is subnormal (below 1.2e-38: a softmax probability of e^-87 times a loss
    gradient) that power is above 2^127: the scale became inf and inf x 0 = NaN reached d feat -- and from there the grid --
    until round 6.  Such samples must come out finite and tiny, their neighbours in the tile untouched, and the weight
    gradients (which replay the scale, csrc/wgrad16.hip) must equal the run with those samples' gradients set to zero to
    fp32 class."""
    torch.manual_seed(11)
    M = 4096
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    ws = dec.ordered_parameters()
    pk = ops.decoder_pack16(ws, precision=precision)
    x = torch.rand(M, 3, device=dev)
    feat = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
    out, saved = ops.decoder_fwd(None, feat, _lib.FEAT_AOS, x, None, M, save="lean", precision=precision, packed16=pk)
    dout = torch.randn(M, 10, device=dev) * 1e-6
    tiny = torch.zeros(M, dtype=torch.bool, device=dev)
    tiny[5::17] = True
    dout_t = dout.clone()
    dout_t[tiny] = torch.randn(int(tiny.sum()), 10, device=dev) * 1e-41          # subnormal rows (incl. exact zeros in places)
    dout_z = dout.clone()
    dout_z[tiny] = 0.0
    res = {}
    for tag, d in (("tiny", dout_t), ("zero", dout_z)):
        g = [torch.zeros_like(w) for w in ws]
        df, dx, _ = ops.decoder_bwd(None, feat, _lib.FEAT_AOS, x, None, out, d, saved, g, M, precision=precision, packed16=pk)
        res[tag] = (df, dx, g)
        assert torch.isfinite(df).all() and torch.isfinite(dx).all() and all(torch.isfinite(t).all() for t in g), tag
    df_t, dx_t, g_t = res["tiny"]
    df_z, dx_z, g_z = res["zero"]
    assert torch.equal(df_t[~tiny], df_z[~tiny]) and torch.equal(dx_t[~tiny], dx_z[~tiny]), "the other samples of the tiles"
    assert float(df_t[tiny].abs().max()) < 1e-30 and float(dx_t[tiny].abs().max()) < 1e-30
    for k, a, b in zip(ops.DECODER_PARAM_ORDER, g_t, g_z):
        assert_close(a, b, 1e-6, "weight gradients beside subnormal samples, grad " + k)


def test_decoder_true_error_of_every_arithmetic_against_fp64(dev):
    """What each decoder arithmetic is worth against the TRUTH (fp64 torch on the host, autograd for the backward) on
    realistic magnitudes: weights as initialised / trained, hash-grid features from tcnn's initial 1e-4 up to 0.2,
    loss gradients of 1e-7 .. 1e-3.  The split-precision f16 path ("f16x3", the default) must be fp32-class -- within
    4x of the fp32-MFMA kernel's own error (measured 1-2.6x: operands carry 22-23 significant bits instead of 24) --
    forward and backward; "bf16x6" carries the fp32 operands exactly and must sit where the fp32-MFMA kernels sit (within
    1.5x of their error either way); plain "f16" is held to its stated 2e-4."""
    M = 20000
    for feat_scale in (1e-4, 0.2):
        torch.manual_seed(0)
        dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
        with torch.no_grad():
            dec.sdf_linear[2].weight.mul_(3.0)
        ws = dec.ordered_parameters()
        packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
        packed_bf = ops.decoder_pack16(ws, precision="bf16x6")
        x = torch.rand(M, 3, device=dev)
        feat = ((torch.rand(M, 32, device=dev) * 2 - 1) * feat_scale).contiguous()
        dout = torch.randn(M, 10, device=dev) * torch.exp(torch.empty(M, 1, device=dev).uniform_(-16.0, -6.0))
        w64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in dec.state_dict().items()}
        x64, f64 = x.cpu().double().requires_grad_(True), feat.cpu().double().requires_grad_(True)
        # A ReLU whose pre-activation sits within rounding noise of zero may fall either way in ANY finite arithmetic (one
        # of 5 M decisions per launch does, for the fp32 kernels as for the others), and with loss gradients spread over four
        # decades one such sample can carry 1e-3 of a gradient's norm: luck, not arithmetic.  Those samples (pre-activation
        # of a hidden unit below 1e-5 in the fp64 evaluation, ~0.3 % of the batch) get no incoming gradient here.
        with torch.no_grad():
            e64 = torch.cat([x64, tcnn_cpu.frequency_forward(x64, 8)], -1)
            h1 = e64 @ w64["pts_linear.0.weight"].T + w64["pts_linear.0.bias"]
            h2 = torch.relu(h1) @ w64["pts_linear.2.weight"].T + w64["pts_linear.2.bias"]
            g3 = torch.cat([h2[:, :64], f64], -1) @ w64["sdf_linear.0.weight"].T + w64["sdf_linear.0.bias"]
            risky = (h1.abs().min(1).values < 1e-5) | (g3.abs().min(1).values < 1e-5)
            dout[risky.to(dev)] = 0.0
            print(f"  {int(risky.sum())} of {M} samples sit on a ReLU threshold")
        ref = path_cpu.decoder_forward(w64, f64, tcnn_cpu.frequency_forward(x64, 8), x64)
        ref.backward(dout.cpu().double())
        err = {}
        for prec in ("f32", "f16x3", "bf16x6", "f16"):
            kw = {} if prec == "f32" else dict(precision=prec, packed16=packed_bf if prec == "bf16x6" else packed16)
            out, saved = ops.decoder_fwd(packed, feat, _lib.FEAT_AOS, x, None, M, save=prec != "f16", **kw)
            e = {"fwd": float((out.cpu().double() - ref.detach()).abs().max())}
            if prec != "f16":
                g = [torch.zeros_like(w) for w in ws]
                dfeat, dx, _ = ops.decoder_bwd(packed, feat, _lib.FEAT_AOS, x, None, out, dout, saved, g, M, **kw)
                rl2 = lambda a, b: float((a.cpu().double() - b).norm() / b.norm())   # noqa: E731
                e.update(dfeat=rl2(dfeat, f64.grad), dx=rl2(dx, x64.grad), w_pts0=rl2(g[0], w64["pts_linear.0.weight"].grad),
                         w_sdf0=rl2(g[6], w64["sdf_linear.0.weight"].grad))
            err[prec] = e
        print(f"features ~{feat_scale:g}: " + "; ".join(f"{p}: " + " ".join(f"{k} {v:.1e}" for k, v in e.items())
                                                      for p, e in err.items()))
        for k in err["f32"]:
            assert err["f16x3"][k] <= 4.0 * err["f32"][k] + 1e-9, f"f16x3 {k}: {err['f16x3'][k]:.2e} vs fp32 {err['f32'][k]:.2e}"
            assert err["bf16x6"][k] <= 1.5 * err["f32"][k] + 1e-9, f"bf16x6 {k}: {err['bf16x6'][k]:.2e} vs fp32 {err['f32'][k]:.2e}"
        assert err["f16"]["fwd"] < 2e-4


@pytest.mark.gpu
def test_random_optimizer_graph_capture_equals_eager_rounds(dev):
    """RandomOptimizer.capture / optimize_graphed (all rounds of a frame as ONE hipGraph replay, inputs through two
    small uploads) must track exactly the pose the eager `optimize` tracks, frame after frame, while the map's
    parameters change in place underneath the recorded graph."""
    from mipsfusion_amd.graph import work_stream
    g = load_golden("ro.npz")
    cfg, ro = _ro_setup(g, dev)
    ro.pre_sampled_particle = T(g["pst"]).to(dev).contiguous()
    m = make_scene(g, cfg, dev).eval()
    stream = torch.cuda.Stream()
    depth, init = T(g["depth"]), T(g["init_pose"])
    with torch.cuda.stream(stream):
        ro.capture(m, 5, stream)
        for trial in range(3):
            if trial:       # the optimiser moves the map in place between frames
                with torch.no_grad():
                    m.embed_fn.params.mul_(1.01)
                    m.decoder.sdf_linear[2].weight.mul_(0.99)
            start = init.clone()
            start[:3, 3] += 0.004 * trial
            eager = ro.optimize(m, depth, start.clone(), None, n_iter=5).cpu()
            graphed = ro.optimize_graphed(depth.to(dev).reshape(-1), start.clone())
            assert torch.equal(eager, graphed), f"frame {trial}: captured rounds differ from eager rounds"
    torch.cuda.synchronize()


def test_operand_buffers_of_the_other_family_are_refused_and_huge_queries_are_cut(dev, monkeypatch):
    """(1) A packed16 buffer of the f16 family must not reach the bf16x6 kernels (they would read it past its end) nor the
    other way round: the buffers differ in size, which -- unlike a python attribute -- survives clone() / views; ops refuses,
    and the library refuses again (packed16_floats of its argument blocks).  (2) bf16x6 takes at most 2^24 - 1 samples per
    launch: forward-only queries beyond that (a 256^3 mesher grid) are cut into several launches -- same values (exercised
    here with the limits lowered)."""
    import ctypes as C
    from mipsfusion_amd._lib import dptr, lib, stream_ptr
    from mipsfusion_amd.model import scene_rep
    torch.manual_seed(11)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    ws = dec.ordered_parameters()
    M = 100
    x = torch.rand(M, 3, device=dev)
    feat = torch.randn(16, M, 2, device=dev) * 0.1
    p_f16, p_bf = ops.decoder_pack16(ws, precision="f16x3"), ops.decoder_pack16(ws, precision="bf16x6")
    for buf, prec in ((p_f16.clone(), "bf16x6"), (p_bf.clone(), "f16x3"), (p_bf[: p_f16.numel() + 8].clone(), "bf16x6")):
        with pytest.raises(RuntimeError, match="packed for the other family|holds"):
            ops.decoder_fwd(None, feat, _lib.FEAT_LEVEL_MAJOR, x, None, M, save=False, precision=prec, packed16=buf)
    out = torch.empty(M, 10, device=dev)
    a = _lib.DecoderFwd16Args.new(M=M, packed16=dptr(p_f16), feat=dptr(feat), x=dptr(x), out=dptr(out),
                                  feat_layout=_lib.FEAT_LEVEL_MAJOR, precision=_lib.PREC["bf16x6"], packed16_floats=p_f16.numel())
    assert lib().mipsf_decoder_fwd16(C.byref(a), stream_ptr()) != 0 and b"other family" in lib().mipsf_last_error()
    a.packed16, a.packed16_floats = dptr(p_bf), p_bf.numel()
    assert lib().mipsf_decoder_fwd16(C.byref(a), stream_ptr()) == 0
    # (2)
    g = load_golden("scene_cfg1.npz")
    m = make_scene(g, cfg_for("scene_cfg1.npz"), dev).eval()
    pts = torch.rand(1000, 3, device=dev)
    with torch.no_grad():
        whole, sdf_whole = m.query_color_sdf(pts), m.query_sdf(pts)
        monkeypatch.setattr(scene_rep, "_MAX_QUERY", 256)
        monkeypatch.setattr(scene_rep, "_QUERY_CHUNK", 192)
        cut, sdf_cut = m.query_color_sdf(pts), m.query_sdf(pts)
    assert torch.equal(whole, cut) and torch.equal(sdf_whole, sdf_cut) and cut.shape == (1000, 10) and sdf_cut.shape == (1000, 1)
