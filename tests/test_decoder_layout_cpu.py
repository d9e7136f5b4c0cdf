"""CPU: lane-level emulation of the register-chained MFMA decoder kernels.

The weight images come from the REAL packing code of the library (mipsf_decoder_pack_host runs the same
``packed_value`` function the GPU pack kernel runs); the MFMA instruction, the accumulator-as-B-operand
chaining, the LDS transposes of the weight-gradient kernel and the natural-layout flush are emulated with
numpy following mipsfusion_amd/csrc/decoder.hip step by step.  The result must equal the oracle decoder
(model/decoder.py:53-75 restated in oracle/path_cpu.py) and its autograd gradients, which validates every
index map of decoder_layout.h without a GPU.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from mipsfusion_amd import _lib
from oracle import path_cpu

HID, N_E, E_SLOTS = 128, 51, 26
PI_F = float(np.float32(np.pi))
HALF_PI_F = float(np.float32(np.pi) / np.float32(2))

LANE = np.arange(64)
J, H = LANE & 31, LANE >> 5


def rowmap(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def feat_of(q, r, h):
    return 32 * q + rowmap(r, h)


def eidx(t, h):
    if t < 24:
        return 3 + (t >> 3) * 16 + 2 * (t & 7) + h
    if t == 24:
        return h
    if t == 25:
        return 2 if h == 0 else -1
    return -1


def mfma(a, b, c):
    """v_mfma_f32_32x32x2_f32: a,b [64] lane vectors, c [16,64] accumulator registers."""
    A = np.zeros((32, 2))
    B = np.zeros((2, 32))
    A[J, H] = a
    B[H, J] = b
    D = A @ B
    out = c.copy()
    for r in range(16):
        out[r] += D[rowmap(r, H), J]
    return out


OFF = {}


def layout_offsets():
    def img(rt, T):
        return rt * T * 64
    o = 0
    for name, rt, T in (("F1", 4, 28), ("F2", 4, 64), ("F3", 4, 48), ("B3", 3, 64), ("B2", 4, 64), ("B1", 2, 64)):
        OFF[name] = (o, rt, T)
        o += img(rt, T)
    OFF["TRGB"] = o
    o += 2 * 58 * 4
    OFF["TS2"] = o
    o += 2 * 64 * 8
    OFF["BIAS"] = o
    o += 3 * 64 * 2
    OFF["BSMALL"] = o
    o += 12
    return o


TOTAL = layout_offsets()


def pack_host(w):
    lib = _lib.lib()
    assert _lib.buffer_size(_lib.SIZE_DECODER_PACKED) == TOTAL
    keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}
    st = _lib.DecoderWeights()
    for field, key in (("w_pts0", "pts_linear.0.weight"), ("b_pts0", "pts_linear.0.bias"),
                       ("w_pts2", "pts_linear.2.weight"), ("b_pts2", "pts_linear.2.bias"),
                       ("w_rgb0", "rgb_linear.0.weight"), ("b_rgb0", "rgb_linear.0.bias"),
                       ("w_sdf0", "sdf_linear.0.weight"), ("b_sdf0", "sdf_linear.0.bias"),
                       ("w_sdf2", "sdf_linear.2.weight"), ("b_sdf2", "sdf_linear.2.bias")):
        setattr(st, field, keep[key].ctypes.data)
    out = np.zeros(TOTAL, dtype=np.float32)
    _lib.check(lib.mipsf_decoder_pack_host(C.byref(st), out.ctypes.data), "pack_host")
    return out.astype(np.float64)


def img_operand(packed, name, rt, t):
    off, _, T = OFF[name]
    idx = off + ((rt * (T // 4) + (t >> 2)) * 64 + LANE) * 4 + (t & 3)
    return packed[idx]


def mfma_layer(packed, name, acc, bfn):
    _, RT, T = OFF[name]
    for t in range(T):
        b = bfn(t)
        for rt in range(RT):
            acc[rt] = mfma(img_operand(packed, name, rt, t), b, acc[rt])
    return acc


def bias(packed, layer):
    acc = np.zeros((4, 16, 64))
    for rt in range(4):
        for r in range(16):
            acc[rt, r] = packed[OFF["BIAS"] + ((layer * 64 + rt * 16 + r) << 1) + H]
    return acc


def load_e(x, tile):
    """x [32,3] of this wave tile -> ev [26,64] exactly as load_e<true> does."""
    ev = np.zeros((E_SLOTS, 64))
    xs = x[J]
    for d in range(3):
        for k in range(8):
            arg = np.float32(np.float32(xs[:, d]) * np.float32(2.0 ** k)).astype(np.float64) * PI_F + np.where(H == 1, HALF_PI_F, 0.0)
            ev[d * 8 + k] = np.sin(np.float32(arg).astype(np.float64))
    ev[24] = np.where(H == 1, xs[:, 1], xs[:, 0])
    ev[25] = np.where(H == 1, 0.0, xs[:, 2])
    return ev


def trgb(packed, slot):       # -> [64 lanes, 4]
    idx = OFF["TRGB"] + ((H * 58 + slot) * 4)[:, None] + np.arange(4)[None]
    return packed[idx]


def ts2(packed, slot):        # -> [64 lanes, 8]
    idx = OFF["TS2"] + ((H * 64 + slot) * 8)[:, None] + np.arange(8)[None]
    return packed[idx]


def swap32(v):
    return v[LANE ^ 32]


def emu_forward(packed, x, feat):
    """x [32,3], feat [32,32] (AoS) -> out [32,10], saved (H1,H2,H3 accumulator images), ev."""
    ev = load_e(x, 0)
    H1 = mfma_layer(packed, "F1", bias(packed, 0), lambda t: ev[t] if t < E_SLOTS else np.zeros(64))
    H1 = np.maximum(H1, 0.0)
    H2 = mfma_layer(packed, "F2", bias(packed, 1), lambda t: H1[t >> 4, t & 15])
    pr = np.zeros((3, 64))
    for slot in range(32):
        w = trgb(packed, slot)
        for c in range(3):
            pr[c] += w[:, c] * H2[2 + (slot >> 4), slot & 15]
    for t in range(E_SLOTS):
        w = trgb(packed, 32 + t)
        for c in range(3):
            pr[c] += w[:, c] * ev[t]
    rgb = np.stack([pr[c] + swap32(pr[c]) + packed[OFF["BSMALL"] + c] for c in range(3)])
    gf = np.stack([feat[J, 2 * u + H] for u in range(16)])
    H3 = mfma_layer(packed, "F3", bias(packed, 2), lambda t: H2[t >> 4, t & 15] if t < 32 else gf[t - 32])
    H3 = np.maximum(H3, 0.0)
    pl = np.zeros((5, 64))
    for slot in range(64):
        w = ts2(packed, slot)
        for c in range(5):
            pl[c] += w[:, c] * H3[slot >> 4, slot & 15]
    lg = np.stack([pl[c] + swap32(pl[c]) + packed[OFF["BSMALL"] + 4 + c] for c in range(5)])
    p = np.exp(lg - lg.max(0))
    p = p / p.sum(0)
    ent = -(p * np.log2(p + 1e-5)).sum(0)
    sdf = ((p * np.arange(5)[:, None]).sum(0) / 4.0 - 0.5) * 2.0
    out = np.zeros((32, 10))
    lo = H == 0
    out[J[lo], 0:3] = rgb[:, lo].T
    out[J[lo], 3] = sdf[lo]
    out[J[lo], 4] = ent[lo]
    out[J[~lo], 5:10] = p[:, ~lo].T
    return out, (H1, H2, H3), ev


def emu_backward(packed, x, out, dout, saved, ev):
    H1, H2, H3 = saved
    p = out[J, 5:10].T
    g = dout[J].T                                   # [10, 64]
    q = p + 1e-5
    dent = -(np.log2(q) + p / (q * np.log(2.0)))
    dp = g[5:10] + g[3] * (0.5 * np.arange(5)[:, None]) + g[4] * dent
    dot = (p * dp).sum(0)
    dlg = p * (dp - dot)
    drgb = g[0:3]
    dG3 = np.zeros((4, 16, 64))
    for slot in range(64):
        w = ts2(packed, slot)
        v = sum(w[:, c] * dlg[c] for c in range(5))
        dG3[slot >> 4, slot & 15] = np.where(H3[slot >> 4, slot & 15] > 0, v, 0.0)
    dIn3 = mfma_layer(packed, "B3", np.zeros((3, 16, 64)), lambda t: dG3[t >> 4, t & 15])
    dfeat = np.zeros((32, 32))
    for r in range(16):
        dfeat[J, rowmap(r, H)] = dIn3[2, r]
    dH2 = np.zeros((4, 16, 64))
    dH2[0], dH2[1] = dIn3[0], dIn3[1]
    de = np.zeros((E_SLOTS, 64))
    for slot in range(32):
        w = trgb(packed, slot)
        dH2[2 + (slot >> 4), slot & 15] = sum(w[:, c] * drgb[c] for c in range(3))
    for t in range(E_SLOTS):
        w = trgb(packed, 32 + t)
        de[t] = sum(w[:, c] * drgb[c] for c in range(3))
    dG1 = mfma_layer(packed, "B2", np.zeros((4, 16, 64)), lambda t: dH2[t >> 4, t & 15])
    dG1 = np.where(H1 > 0, dG1, 0.0)
    dE = mfma_layer(packed, "B1", np.zeros((2, 16, 64)), lambda t: dG1[t >> 4, t & 15])
    for t in range(E_SLOTS):
        de[t] += dE[t >> 4, t & 15]
    # PE chain
    xs = x[J]
    gx = np.zeros((3, 64))
    for d in range(3):
        for k in range(8):
            arg = np.float32(np.float32(np.float32(xs[:, d]) * np.float32(2.0 ** k)).astype(np.float64) * PI_F
                             + np.where(H == 1, HALF_PI_F, 0.0)).astype(np.float64)
            gx[d] += de[d * 8 + k] * ((2.0 ** k) * PI_F * np.cos(arg))
    gx[0] += np.where(H == 0, de[24], 0.0)
    gx[1] += np.where(H == 1, de[24], 0.0)
    gx[2] += np.where(H == 0, de[25], 0.0)
    gx = gx + gx[:, LANE ^ 32]
    dx = gx[:, :32].T
    # module-mode outputs: d embed_pos and direct d x
    dpe = np.zeros((32, 48))
    for t in range(24):
        dpe[J, (t >> 3) * 16 + 2 * (t & 7) + H] = de[t]
    return dict(dfeat=dfeat, dx=dx, dpe=dpe, dG1=dG1, dH2=dH2, dG3=dG3, dlg=dlg, drgb=drgb, de=de)


def emu_wgrad(tiles):
    """tiles: list (<=4 wave tiles) of dicts with saved, ev, feat, bwd -> natural-layout gradients."""
    LDW = 129

    def stage(dst, m, w, row_shift=0, rts=range(4)):
        for rt in rts:
            for r in range(16):
                dst[feat_of(rt, r, H) + row_shift, 32 * w + J] = m[rt, r]

    def mma(XT, YT, rtile, ctile, first8=False):
        acc = np.zeros((16, 64))
        for t in range(64):
            a = XT[32 * rtile + J, 2 * t + H]
            if first8:
                a = np.where(J < 8, a, 0.0)
            b = YT[32 * ctile + J, 2 * t + H]
            acc = mfma(a, b, acc)
        return acc

    def flush(G, acc, rtile, ctile, out_dim, in_dim):
        for r in range(16):
            rows = 32 * rtile + rowmap(r, H)
            cols = 32 * ctile + J
            ok = (rows < out_dim) & (cols < in_dim)
            G[rows[ok], cols[ok]] = acc[r][ok]

    XT = np.full((128, LDW), np.nan)
    YT = np.full((128, LDW), np.nan)
    nw = len(tiles)
    g = {}

    def staged(fx, fy):
        XT[:] = np.nan
        YT[:] = np.nan
        for w in range(4):
            if w < nw:
                fx(w, tiles[w])
                fy(w, tiles[w])
            else:
                XT[:, 32 * w:32 * w + 32] = 0.0
                YT[:, 32 * w:32 * w + 32] = 0.0

    # sdf0
    def y_sdf0(w, tl):
        stage(YT, tl["saved"][1], w, rts=range(2))
        for u in range(16):
            YT[64 + 2 * u + H, 32 * w + J] = tl["feat"][J, 2 * u + H]
    staged(lambda w, tl: stage(XT, tl["bwd"]["dG3"], w), y_sdf0)
    G = np.zeros((128, 96))
    for w in range(4):
        for ct in range(3):
            flush(G, mma(XT, YT, w, ct), w, ct, 128, 96)
    g["sdf_linear.0.weight"], g["sdf_linear.0.bias"] = G, XT[:, :128].sum(1)
    # pts2
    staged(lambda w, tl: stage(XT, tl["bwd"]["dH2"], w), lambda w, tl: stage(YT, tl["saved"][0], w))
    G = np.zeros((128, 128))
    for w in range(4):
        for ct in range(4):
            flush(G, mma(XT, YT, w, ct), w, ct, 128, 128)
    g["pts_linear.2.weight"], g["pts_linear.2.bias"] = G, XT[:, :128].sum(1)
    # pts0
    def y_e(w, tl, shift=0):
        for t in range(E_SLOTS):
            for hh in (0, 1):
                e = eidx(t, hh)
                if e >= 0:
                    sel = H == hh
                    YT[shift + e, 32 * w + J[sel]] = tl["ev"][t][sel]
    def y_pts0(w, tl):
        y_e(w, tl)
        YT[51:64, :128] = 0.0
    staged(lambda w, tl: stage(XT, tl["bwd"]["dG1"], w), y_pts0)
    YT[51:64, :128] = 0.0
    G = np.zeros((128, 51))
    for w in range(4):
        for ct in range(2):
            flush(G, mma(XT, YT, w, ct), w, ct, 128, 51)
    g["pts_linear.0.weight"], g["pts_linear.0.bias"] = G, XT[:, :128].sum(1)
    # sdf2 / rgb0: X = 8 rows
    def x_small(w, tl):
        XT[0:5, 32 * w:32 * w + 32] = tl["bwd"]["dlg"][:, :32]
        XT[5:8, 32 * w:32 * w + 32] = tl["bwd"]["drgb"][:, :32]
    staged(x_small, lambda w, tl: stage(YT, tl["saved"][2], w))
    Gs2 = np.zeros((5, 128))
    for w in range(4):
        acc = mma(XT, YT, 0, w, first8=True)
        for r in range(16):
            rows, cols = rowmap(r, H), 32 * w + J
            ok = rows < 5
            Gs2[rows[ok], cols[ok]] = acc[r][ok]
    g["sdf_linear.2.weight"], g["sdf_linear.2.bias"] = Gs2, XT[0:5, :128].sum(1)
    g["rgb_linear.0.bias"] = XT[5:8, :128].sum(1)
    def y_rgb(w, tl):
        stage(YT, tl["saved"][1], w, row_shift=-64, rts=range(2, 4))
        y_e(w, tl, shift=64)
        YT[115:128, :128] = 0.0
    staged(x_small, y_rgb)
    YT[115:128, :128] = 0.0
    Grgb = np.zeros((3, 115))
    for w in range(4):
        acc = mma(XT, YT, 0, w, first8=True)
        for r in range(16):
            rows, cols = rowmap(r, H), 32 * w + J
            ok = (rows >= 5) & (rows < 8) & (cols < 115)
            Grgb[rows[ok] - 5, cols[ok]] = acc[r][ok]
    g["rgb_linear.0.weight"] = Grgb
    return g


@pytest.fixture(scope="module")
def problem():
    torch.manual_seed(3)
    cfg = {"grid": {"hash_size": 10}, "pos": {"n_bins": 8}, "training": {"norm_factor": 1.0}}
    scene = path_cpu.CpuScene(cfg, [[-1, 1]] * 3, [7, 7, 7]).double()
    w = {k: v.detach() for k, v in scene.decoder_weights().items()}
    n = 96                                           # 3 wave tiles of one block tile
    x = torch.rand(n, 3, dtype=torch.float64)
    feat = torch.randn(n, 32, dtype=torch.float64) * 0.3
    return scene, w, x, feat


def oracle_decoder(w, x, feat, dout):
    from oracle import tcnn_cpu
    x = x.clone().requires_grad_(True)
    feat = feat.clone().requires_grad_(True)
    ws = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    # same fp32-argument frequency encoding as the oracle, in float64 arithmetic afterwards
    pe_cols = []
    for d in range(3):
        for k in range(8):
            for s in (0, 1):
                t = (x[:, d].float() * float(2 ** k)).double()
                arg32 = (t * PI_F + s * HALF_PI_F).float().double()
                # differentiable: value sin(arg32), derivative 2^k*pi*cos(arg32)
                pe_cols.append(_SinAt.apply(x[:, d], arg32, float(2 ** k) * PI_F))
    pe = torch.stack(pe_cols, -1)
    out = path_cpu.decoder_forward(ws, feat, pe, x)
    out.backward(dout)
    return out.detach(), x.grad, feat.grad, {k: v.grad for k, v in ws.items()}


class _SinAt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xd, arg, scale):
        ctx.save_for_backward(arg)
        ctx.scale = scale
        return torch.sin(arg)

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return g * ctx.scale * torch.cos(arg), None, None


def test_packed_size_matches_library():
    assert _lib.buffer_size(_lib.SIZE_DECODER_PACKED) == TOTAL


def test_chain_forward_backward_and_wgrad_match_oracle(problem):
    scene, w, x, feat = problem
    packed = pack_host({k: v.float().numpy() for k, v in w.items()})
    w64 = {k: v.float().double() for k, v in w.items()}        # the values the packer saw
    torch.manual_seed(5)
    dout = torch.randn(x.shape[0], 10, dtype=torch.float64)
    ref_out, ref_dx, ref_dfeat, ref_gw = oracle_decoder(w64, x, feat, dout)

    tiles, outs, dxs, dfeats = [], [], [], []
    for t in range(3):
        sl = slice(32 * t, 32 * t + 32)
        xt, ft = x[sl].numpy(), feat[sl].numpy()
        out, saved, ev = emu_forward(packed, xt, ft)
        bwd = emu_backward(packed, xt, out, dout[sl].numpy(), saved, ev)
        tiles.append(dict(saved=saved, ev=ev, feat=ft, bwd=bwd))
        outs.append(out), dxs.append(bwd["dx"]), dfeats.append(bwd["dfeat"])
    np.testing.assert_allclose(np.concatenate(outs), ref_out.numpy(), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(np.concatenate(dfeats), ref_dfeat.numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(np.concatenate(dxs), ref_dx.numpy(), rtol=1e-8, atol=1e-9)
    g = emu_wgrad(tiles)
    for k, v in ref_gw.items():
        np.testing.assert_allclose(g[k], v.numpy(), rtol=1e-8, atol=1e-9, err_msg=k)


def test_module_mode_embed_pos_gradient_layout(problem):
    """pe_mode 1: d embed_pos index (t>>3)*16 + 2*(t&7) + h and the direct x slots."""
    scene, w, x, feat = problem
    packed = pack_host({k: v.float().numpy() for k, v in w.items()})
    w64 = {k: v.float().double() for k, v in w.items()}
    xt, ft = x[:32], feat[:32]
    pe = torch.randn(32, 48, dtype=torch.float64).requires_grad_(True)
    xx = xt.clone().requires_grad_(True)
    out = path_cpu.decoder_forward(w64, ft, pe, xx)
    dout = torch.randn(32, 10, dtype=torch.float64)
    out.backward(dout)
    # emulate with externally supplied e values
    ev = np.zeros((E_SLOTS, 64))
    for t in range(24):
        ev[t] = pe.detach().numpy()[J, (t >> 3) * 16 + 2 * (t & 7) + H]
    ev[24] = np.where(H == 1, xt.numpy()[J, 1], xt.numpy()[J, 0])
    ev[25] = np.where(H == 1, 0.0, xt.numpy()[J, 2])
    global load_e
    orig = load_e
    try:
        load_e = lambda x_, tile: ev          # noqa: E731
        o, saved, _ = emu_forward(packed, xt.numpy(), ft.numpy())
    finally:
        load_e = orig
    np.testing.assert_allclose(o, out.detach().numpy(), rtol=1e-9, atol=1e-10)
    bwd = emu_backward(packed, xt.numpy(), o, dout.numpy(), saved, ev)
    np.testing.assert_allclose(bwd["dpe"], pe.grad.numpy(), rtol=1e-8, atol=1e-10)
    de = bwd["de"]
    np.testing.assert_allclose(de[24][:32], xx.grad.numpy()[:, 0], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(de[24][32:], xx.grad.numpy()[:, 1], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(de[25][:32], xx.grad.numpy()[:, 2], rtol=1e-8, atol=1e-10)


# ----------------------------------------------------------------------------------------------------------------------
# The LDS image of the transpose-read weight-gradient kernel (csrc/wgrad16.hip, w16t_addr; round 6), emulated lane by lane:
# a 32 x 32 block in the records' load layout is written plane by plane with ds_write_b128 and read back with
# ds_read_b64_tr_b16 as an MFMA operand.  The instruction's lane mapping is the one tools/micro/tr_probe.hip checks on the
# device; the bank rules are MI355X_MICROARCH.md's (ds_write_b128: 8 contiguous lanes per LDS cycle, banks of 4 bytes modulo
# 32; transposing read: 32 lanes per cycle, banks modulo 64).  Checked: the operand's elements in the order the READY operand
# planes use (accumulator rows: k-step m, half kg, element u = sample 16 m + 8 (u >> 2) + 4 kg + (u & 3)) for both operand
# widths, and no bank conflict in any of the three access kinds.
def _w16t_addr(lane):
    j, h = lane & 31, lane >> 5
    sw = ((j >> 1) & 1) | ((((j >> 2) ^ (j >> 3)) & 1) << 1)
    wr0, wr1 = j * 64 + ((h ^ sw) * 16), j * 64 + (((2 + h) ^ sw) * 16)
    g, ii = lane >> 4, lane & 15
    hh, aa, s1 = ii & 1, (ii >> 1) & 1, (ii >> 3) & 1
    q, kg = g & 1, g >> 1
    row0, lo = 4 * kg + (ii >> 2), (hh ^ s1) * 16 + 8 * aa
    rd0 = row0 * 64 + 32 * (q ^ kg) + lo
    rd1 = (row0 + 8) * 64 + 32 * (q ^ 1 ^ kg) + lo
    base = (16 * (g >> 1) + 4 * (g & 1) + (ii >> 2)) * 64 + (hh ^ s1) * 16 + 8 * aa
    return dict(wr0=wr0, wr1=wr1, rd0=rd0, rd1=rd1, rsa=base + 32 * (g & 1), rsb=base + 32 * (1 - (g & 1)))


def _tr_read(lds, addrs):
    """ds_read_b64_tr_b16: result lane i of a 16-lane group, element k = element (i & 3) of the chunk source lane 4 k + (i >> 2) addresses"""
    out = np.zeros((64, 4), dtype=lds.dtype)
    for lane in range(64):
        grp, i = lane & ~15, lane & 15
        for k in range(4):
            out[lane, k] = lds[addrs[grp + 4 * k + (i >> 2)] // 2 + (i & 3)]
    return out


def _max_bank_load(addrs, lanes, n_bytes, n_banks):
    load = {}
    for lane in lanes:
        for d in range(n_bytes // 4):
            b = (addrs[lane] // 4 + d) % n_banks
            load[b] = load.get(b, 0) + 1
    return max(load.values())


def test_wgrad_transpose_read_lds_layout_emulation():
    V = (np.arange(32 * 32).reshape(32, 32) + 1).astype(np.int64)          # V[sample][feature]
    ad = [_w16t_addr(lane) for lane in range(64)]
    lds = np.zeros(1024, dtype=np.int64)                                    # one plane: 2 KB of 16-bit elements
    for lane in range(64):
        j, h = lane & 31, lane >> 5
        for q in range(2):
            for u in range(8):                                              # load layout: feature 16 q + 8 (u >> 2) + 4 h + (u & 3)
                lds[ad[lane]["wr%d" % q] // 2 + u] = V[j, 16 * q + 8 * (u >> 2) + 4 * h + (u & 3)]
    for m in range(2):                                                      # 32-wide operand (v_mfma_f32_32x32x16)
        X = np.concatenate([_tr_read(lds, [a["rd0"] + 1024 * m for a in ad]), _tr_read(lds, [a["rd1"] + 1024 * m for a in ad])], 1)
        for lane in range(64):
            for u in range(8):
                assert X[lane, u] == V[16 * m + 8 * (u >> 2) + 4 * (lane >> 5) + (u & 3), lane & 31]
    for cg in range(2):                                                     # 16-wide operand (v_mfma_f32_16x16x32)
        a0 = [a["rsa" if cg == 0 else "rsb"] for a in ad]
        a1 = [a["rsb" if cg == 0 else "rsa"] + 512 for a in ad]
        Y = np.concatenate([_tr_read(lds, a0), _tr_read(lds, a1)], 1)
        for lane in range(64):
            g = lane >> 4
            for u in range(8):
                assert Y[lane, u] == V[16 * (g >> 1) + 8 * (u >> 2) + 4 * (g & 1) + (u & 3), 16 * cg + (lane & 15)]
    # bank conflicts: none
    for key in ("wr0", "wr1"):
        for g0 in range(0, 64, 8):
            assert _max_bank_load([a[key] for a in ad], range(g0, g0 + 8), 16, 32) == 1
    for key, off in (("rd0", 0), ("rd1", 0), ("rd0", 1024), ("rsa", 0), ("rsb", 512), ("rsb", 0), ("rsa", 512)):
        for half in (range(0, 32), range(32, 64)):
            assert _max_bank_load([a[key] + off for a in ad], half, 8, 64) == 1
