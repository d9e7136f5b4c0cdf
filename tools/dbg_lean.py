import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(1100)
M = 1000
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed16 = ops.decoder_pack16(ws)
x = torch.rand(M, 3, device=dev); feat = (torch.randn(M, 32, device=dev) * 0.3).contiguous()
lay = _lib.FEAT_AOS
dout = torch.randn(M, 10, device=dev) * 1e-4
def run(save, rc):
    out, saved = ops.decoder_fwd(None, feat, lay, x, None, M, save=save, precision="f16x3", packed16=packed16)
    g = [torch.zeros_like(w) for w in ws]
    ops.decoder_bwd(None, feat, lay, x, None, out, dout, saved, g, M, precision="f16x3", packed16=packed16, wgrad_precision="stream_f16x3", recompute_h1=rc)
    torch.cuda.synchronize()
    return out, saved, g
o1, s1, g1 = run(True, True)
o2, s2, g2 = run(True, True)
o3, s3, g3 = run("lean", True)
o4, s4, g4 = run(True, False)
n_h = s1.numel()
print("out equal", torch.equal(o1, o3))
for k, a, b, c, d in zip(ops.DECODER_PARAM_ORDER, g1, g2, g3, g4):
    print(f"{k:24s} rerun {float((a-b).abs().max()):.1e}  lean {float((a-c).abs().max()):.1e}  stored-H1 {float((a-d).abs().max()):.1e}  max {float(a.abs().max()):.1e}")
# which parts of saved differ (tile record = 12288 floats: mats 0,1,2 of 4096 each)
nt = (M + 31) // 32
rec1 = s1[:((M + 127) // 128) * 4 * 12288].view(-1, 3, 4096)[:nt]
rec3 = s3[:((M + 127) // 128) * 4 * 12288].view(-1, 3, 4096)[:nt]
for m in range(3):
    print("mat", m, "equal", torch.equal(rec1[:, m], rec3[:, m]))
tail1 = s1[((M + 127) // 128) * 4 * 12288:]; tail3 = s3[((M + 127) // 128) * 4 * 12288:]
print("masks equal", torch.equal(tail1.view(torch.int32), tail3.view(torch.int32)))
