"""True (fp64) error of the decoder forward per arithmetic, on realistic inputs (trained-magnitude weights, grid features
from U(-1e-4,1e-4) up to 0.2)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
from oracle import path_cpu, tcnn_cpu
dev = torch.device("cuda:0")
M = 20000
for feat_scale in (1e-4, 0.2):
    torch.manual_seed(0)
    dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
    with torch.no_grad():
        dec.sdf_linear[2].weight.mul_(3.0)
    ws = dec.ordered_parameters()
    packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
    x = torch.rand(M, 3, device=dev)
    feat = ((torch.rand(M, 32, device=dev) * 2 - 1) * feat_scale).contiguous()
    w64 = {k: v.detach().cpu().double() for k, v in dec.state_dict().items()}
    pe32 = tcnn_cpu.frequency_forward(x.cpu(), 8)              # fp32 PE as every path sees it
    ref = path_cpu.decoder_forward(w64, feat.cpu().double(), pe32.double(), x.cpu().double())
    for prec in ("f32", "f16x3", "f16"):
        kw = {} if prec == "f32" else dict(precision=prec, packed16=packed16)
        out, _ = ops.decoder_fwd(packed, feat, _lib.FEAT_AOS, x, None, M, save=False, **kw)
        d = (out.cpu().double() - ref).abs()
        print(f"feat {feat_scale:g} {prec:6s}: max abs err rgb {float(d[:, :3].max()):.2e}  sdf {float(d[:, 3].max()):.2e}  prob {float(d[:, 5:].max()):.2e}"
              f"   rms sdf {float(d[:, 3].pow(2).mean().sqrt()):.2e}")
