import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mipsfusion_amd import _lib, ops
from oracle import tcnn_cpu
dev = torch.device("cuda:0")
PLS = float(2.0 ** (math.log2(16) / 15))
log2_t, M = 19, int(sys.argv[1]) if len(sys.argv) > 1 else 100000
meta = _lib.make_grid_meta(16, 2, log2_t, 16, PLS)
om = tcnn_cpu.make_grid_meta(16, 2, log2_t, 16, PLS)
torch.manual_seed(1)
x = torch.rand(M, 3); params = torch.rand(om.n_params) - 0.5; dy = torch.rand(M, 32) + 0.5
dp_ref, _ = tcnn_cpu.hashgrid_backward(x, params, dy, om, need_dx=False)
dp = torch.zeros(om.n_params, device=dev)
ops.hashgrid_bwd(x.to(dev), params.to(dev), dy.to(dev), dp, meta, _lib.FEAT_AOS, None)
dp = dp.cpu()
offs = om.offsets
for l in range(16):
    a, b = dp[2*offs[l]:2*offs[l+1]].view(-1, 2), dp_ref[2*offs[l]:2*offs[l+1]].view(-1, 2)
    bad = ((a - b).abs().max(1).values > 1e-4 * b.abs().max()).numpy()
    if bad.any():
        idx = np.nonzero(bad)[0]
        # contiguous-ish ranges
        cuts = np.nonzero(np.diff(idx) > 2000)[0]
        starts = np.r_[idx[0], idx[cuts + 1]]; ends = np.r_[idx[cuts], idx[-1]]
        zero = float((a[idx].abs().max(1).values == 0).float().mean())
        print(f"level {l} size {offs[l+1]-offs[l]}: {bad.sum()} bad entries, frac exactly-zero {zero:.2f}, ranges", list(zip(starts.tolist(), ends.tolist()))[:8])
    else:
        print(f"level {l}: ok")
