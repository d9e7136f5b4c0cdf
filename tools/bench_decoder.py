"""GPU microbench: decoder fwd / bwd chain / wgrad at M = 262144 (level-major features, PE in-kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mipsfusion_amd import _lib, ops
if os.environ.get("MIPSF_LIB_VARIANT"):      # private experiment build (tools/micro/build_variant.sh)
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro",
                                 "libmipsf_%s.so" % os.environ["MIPSF_LIB_VARIANT"])
    print("variant", os.environ["MIPSF_LIB_VARIANT"])
from mipsfusion_amd.model import MLP_reg
dev = torch.device("cuda:0")
torch.manual_seed(0)
M = 4096 * 64
dec = MLP_reg({}, input_ch=32, input_ch_pos=48).to(dev)
ws = dec.ordered_parameters()
packed = ops.decoder_pack(ws)
feat = torch.randn(16, M, 2, device=dev) * 0.1
x = torch.rand(M, 3, device=dev)
dout = torch.randn(M, 10, device=dev) * 1e-3
grads = [torch.zeros_like(w) for w in ws]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out, saved = ops.decoder_fwd(packed, feat, _lib.FEAT_LEVEL_MAJOR, x, None, M, True)
t_f = timeit(lambda: ops.decoder_fwd(packed, feat, _lib.FEAT_LEVEL_MAJOR, x, None, M, True))
t_fn = timeit(lambda: ops.decoder_fwd(packed, feat, _lib.FEAT_LEVEL_MAJOR, x, None, M, False))
fl = 72370 * M / 1e6
res = {}
for fused in (False,):
    for g in grads: g.zero_()
    r = ops.decoder_bwd(packed, feat, _lib.FEAT_LEVEL_MAJOR, x, None, out, dout, saved, grads, M)
    res[fused] = (r[0].clone(), r[1].clone(), [g.clone() for g in grads])
    ops.PROFILE = {}
    for _ in range(10): ops.decoder_bwd(packed, feat, _lib.FEAT_LEVEL_MAJOR, x, None, out, dout, saved, grads, M)
    torch.cuda.synchronize()
    p = ops.profile_summary(); ops.PROFILE = None
    print("fused" if fused else "split", {k: round(v[1] * 1e3, 1) for k, v in p.items()})
def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
if os.environ.get("MIPSF_DUMP"):            # A/B of two library builds: dump one, compare the other
    torch.save(res[False], os.environ["MIPSF_DUMP"])
if os.environ.get("MIPSF_COMPARE"):
    ref = torch.load(os.environ["MIPSF_COMPARE"])
    print("vs dump: dfeat %.2e dx %.2e" % (rel(res[False][0], ref[0]), rel(res[False][1], ref[1])),
          "grads", ["%.1e" % rel(a, b) for a, b in zip(res[False][2], ref[2])])
print(f"fwd(save) {t_f:7.1f} us {fl/t_f:6.1f} TF | fwd(nosave) {t_fn:7.1f} us")
