"""The weight-gradient kernel in both arithmetics at M = 262144 (for rocprofv3 runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed16 = ops.decoder_pack16(ws)
M = 262144
x = torch.rand(M, 3, device=dev); feat = torch.randn(16, M, 2, device=dev) * 0.3
L = _lib.FEAT_LEVEL_MAJOR
dout = torch.randn(M, 10, device=dev) * 1e-5
out, saved = ops.decoder_fwd(None, feat, L, x, None, M, save=True, precision="f16x3", packed16=packed16)
for _ in range(5):
    for wp in ("f32", "bf16x3"):
        g = [torch.zeros_like(w) for w in ws]
        ops.decoder_bwd(None, feat, L, x, None, out, dout, saved, g, M, precision="f16x3", packed16=packed16, wgrad_precision=wp)
torch.cuda.synchronize()
