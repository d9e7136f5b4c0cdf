import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mipsfusion_amd import _lib, ops
from oracle import tcnn_cpu
dev = torch.device("cuda:0")
PLS = float(2.0 ** (math.log2(16) / 15))
for log2_t, M in ((19, 96), (19, 5000), (16, 3000)):
    meta = _lib.make_grid_meta(16, 2, log2_t, 16, PLS)
    om = tcnn_cpu.make_grid_meta(16, 2, log2_t, 16, PLS)
    torch.manual_seed(1)
    x = torch.rand(M, 3); params = torch.rand(om.n_params) - 0.5; dy = torch.randn(M, 32)
    dp_ref, _ = tcnn_cpu.hashgrid_backward(x, params, dy, om, need_dx=False)
    dp = torch.zeros(om.n_params, device=dev)
    ops.hashgrid_bwd(x.to(dev), params.to(dev), dy.to(dev), dp, meta, _lib.FEAT_AOS, None)
    dp = dp.cpu()
    print(f"T=2^{log2_t} M={M}: total err {float((dp-dp_ref).abs().max()):.3e}  max {float(dp_ref.abs().max()):.3e}")
    offs = om.offsets
    for l in range(16):
        a, b = dp[2*offs[l]:2*offs[l+1]], dp_ref[2*offs[l]:2*offs[l+1]]
        err = (a - b).abs()
        if float(err.max()) > 1e-5 * float(b.abs().max() + 1e-30):
            bad = (err > 1e-5 * b.abs().max()).nonzero().squeeze(-1)
            e0 = int(bad[0]) // 2
            print(f"  level {l}: size {offs[l+1]-offs[l]} bad floats {bad.numel()} first entry {e0} last {int(bad[-1])//2} gpu {a[bad[0]]:.4f} ref {b[bad[0]]:.4f}; sum gpu {float(a.sum()):.4f} ref {float(b.sum()):.4f}")
