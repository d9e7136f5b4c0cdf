import sys, time, torch
sys.path.insert(0, "/root/repo")
from mipsfusion_amd import ops, _lib
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed, packed16 = ops.decoder_pack(ws), ops.decoder_pack16(ws)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M in (64000, 262144, 768000):
    x = torch.rand(M, 3, device=dev); feat = torch.randn(16, M, 2, device=dev) * 0.3
    L = _lib.FEAT_LEVEL_MAJOR
    out = {}
    for prec in ("f32", "f16x3", "f16"):
        kw = {} if prec == "f32" else dict(precision=prec, packed16=packed16)
        out[prec + " fwd"] = t(lambda: ops.decoder_fwd(packed, feat, L, x, None, M, save=False, **kw))
        if prec != "f16":
            out[prec + " fwd+save"] = t(lambda: ops.decoder_fwd(packed, feat, L, x, None, M, save=True, **kw))
        out[prec + " sdf"] = t(lambda: ops.decoder_fwd_sdf(packed, feat, L, x, None, M, **kw))
    print(M, {k: round(v, 1) for k, v in out.items()})
