"""Decoder kernels of one training step, one arithmetic after the other, on a mapping-like batch (M = 4096 x 64, the second
half of most rays without a gradient: ~0.65 of the 32-sample tiles live): forward (evaluation and training forms), backward
chain, weight gradients -- microseconds per launch (events on the launch stream).  python tools/bench_decoder_modes.py [M]"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from mipsfusion_amd import ops, _lib                 # noqa: E402
from mipsfusion_amd.model import MLP_reg             # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 64
L = _lib.FEAT_LEVEL_MAJOR
x = torch.rand(M, 3, device=dev)
feat = torch.randn(16, M, 2, device=dev) * 0.3
dout = torch.randn(M, 10, device=dev) * 1e-5
ray = torch.arange(M, device=dev) // 64
s_in_ray = torch.arange(M, device=dev) % 64
dead = (s_in_ray >= 32) & (ray % 10 < 7)             # 35 % of the tiles
dout[dead] = 0.0
packs = {"f32": ops.decoder_pack(ws), "f16x3": ops.decoder_pack16(ws), "bf16x6": ops.decoder_pack16(ws, precision="bf16x6")}


def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for prec in ("f32", "f16x3", "bf16x6"):
    pk = packs[prec]
    kw = {} if prec == "f32" else dict(precision=prec, packed16=pk)
    p32 = pk if prec == "f32" else None
    res = {}
    res["fwd"] = t(lambda: ops.decoder_fwd(p32, feat, L, x, None, M, save=False, **kw))
    saves = [True] + (["lean", "masks"] if prec != "f32" else [])
    for sv in saves:
        res[f"fwd save={sv}"] = t(lambda: ops.decoder_fwd(p32, feat, L, x, None, M, save=sv, **kw))
    for sv in ([True] if prec == "f32" else [True, "lean"]):
        out, saved = ops.decoder_fwd(p32, feat, L, x, None, M, save=sv, **kw)
        g = [torch.zeros_like(w) for w in ws]
        def bw():
            if hasattr(saved, "mipsf_tile_live"):
                saved.mipsf_tile_live[1] = False
            ops.decoder_bwd(p32, feat, L, x, None, out, dout, saved, g, M, **kw)
        res[f"bwd total (record {sv})"] = t(bw)
        # per-kernel: the ops layer's own timers
        ops.PROFILE = {}
        for _ in range(10):
            bw()
        torch.cuda.synchronize()
        for k, (n, ms) in ops.profile_summary().items():
            res[f"  {k} (record {sv})"] = ms * 1e3
        ops.PROFILE = None
    print(prec, {k: round(v, 1) for k, v in res.items()})
