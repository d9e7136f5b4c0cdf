import sys, os, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd._lib import lib, dptr, stream_ptr, check
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
M = 1000
torch.manual_seed(1)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
x = torch.rand(M, 3, device=dev); feat = (torch.randn(M, 32, device=dev) * 0.3).contiguous(); L = _lib.FEAT_AOS
out, saved = ops.decoder_fwd(packed, feat, L, x, None, M, save=True)
for scale in (1.0, 1e-6):
    dout = torch.randn(M, 10, device=dev) * scale
    n = lib().mipsf_decoder_dact_floats(M)
    res = {}
    for prec in ("f32", "f16x3"):
        dact = torch.zeros(n, device=dev); dfeat = torch.empty_like(feat); dx = torch.empty(M, 3, device=dev)
        if prec == "f32":
            check(lib().mipsf_decoder_bwd_chain(dptr(packed), L, dptr(x), 0, dptr(out), dptr(dout), dptr(saved), dptr(dfeat), dptr(dx), None, dptr(dact), M, stream_ptr()))
        else:
            check(lib().mipsf_decoder_bwd_chain16(dptr(packed16), L, dptr(x), dptr(out), dptr(dout), dptr(saved), dptr(dfeat), dptr(dx), dptr(dact), M, stream_ptr()))
        res[prec] = (dact, dfeat, dx)
    n_act = ((M + 127) // 128) * 4 * 192 * 64
    a, b = res["f32"][0][:n_act].view(-1, 3, 16, 64, 4), res["f16x3"][0][:n_act].view(-1, 3, 16, 64, 4)
    nt = (M + 31) // 32
    for mat, name in ((0, "dG1"), (1, "dH2"), (2, "dG3")):
        d = (a[:nt, mat] - b[:nt, mat]).abs()
        print(scale, name, "max err", float(d.max()), "scale", float(a[:nt, mat].abs().max()), "bad pieces", sorted(set((d > 1e-4 * float(a[:nt, mat].abs().max())).nonzero()[:, 1].tolist()))[:16])
    print(scale, "dsmall err", float((res["f32"][0][n_act:] - res["f16x3"][0][n_act:]).abs().max()), "dfeat", float((res["f32"][1] - res["f16x3"][1]).abs().max()), "dx", float((res["f32"][2] - res["f16x3"][2]).abs().max()))
