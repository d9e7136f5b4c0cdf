"""One launch group of each f16 decoder forward variant at M = 262144 (for rocprofv3 --pmc / --kernel-trace runs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
x = torch.rand(M, 3, device=dev); feat = torch.randn(16, M, 2, device=dev) * 0.3
L = _lib.FEAT_LEVEL_MAJOR
for _ in range(5):
    for prec in ("f16x3", "f16"):
        ops.decoder_fwd(packed, feat, L, x, None, M, save=False, precision=prec, packed16=packed16)
        ops.decoder_fwd_sdf(packed, feat, L, x, None, M, precision=prec, packed16=packed16)
    ops.decoder_fwd(packed, feat, L, x, None, M, save=True, precision="f16x3", packed16=packed16)
torch.cuda.synchronize()
