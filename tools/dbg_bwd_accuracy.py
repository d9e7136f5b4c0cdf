"""True (fp64 autograd) error of the decoder backward per arithmetic on realistic gradient magnitudes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
from oracle import path_cpu, tcnn_cpu
dev = torch.device("cuda:0")
M = 20000
torch.manual_seed(0)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
with torch.no_grad():
    dec.sdf_linear[2].weight.mul_(3.0)
ws = dec.ordered_parameters()
packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
x = torch.rand(M, 3, device=dev)
feat = ((torch.rand(M, 32, device=dev) * 2 - 1) * 0.2).contiguous()
dout = torch.randn(M, 10, device=dev) * torch.exp(torch.empty(M, 1, device=dev).uniform_(-16.0, -6.0))
# fp64 reference through autograd (PE computed in fp64 from x so that d/dx is exact)
w64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in dec.state_dict().items()}
x64 = x.cpu().double().requires_grad_(True); f64 = feat.cpu().double().requires_grad_(True)
pe = tcnn_cpu.frequency_forward(x64, 8)
ref = path_cpu.decoder_forward(w64, f64, pe, x64)
ref.backward(dout.cpu().double())
for prec in ("f32", "f16x3"):
    kw = {} if prec == "f32" else dict(precision=prec, packed16=packed16)
    out, saved = ops.decoder_fwd(packed, feat, _lib.FEAT_AOS, x, None, M, save=True, **kw)
    g = [torch.zeros_like(w) for w in ws]
    dfeat, dx, _ = ops.decoder_bwd(packed, feat, _lib.FEAT_AOS, x, None, out, dout, saved, g, M, **kw)
    def rel(a, b):
        return float((a.cpu().double() - b).norm() / b.norm()), float((a.cpu().double() - b).abs().max() / b.abs().max())
    print(prec, "dfeat (L2, max)", "%.2e %.2e" % rel(dfeat, f64.grad), " dx", "%.2e %.2e" % rel(dx, x64.grad),
          " g_w_pts0", "%.2e %.2e" % rel(g[0], w64["pts_linear.0.weight"].grad), " g_w_sdf0", "%.2e %.2e" % rel(g[6], w64["sdf_linear.0.weight"].grad))
