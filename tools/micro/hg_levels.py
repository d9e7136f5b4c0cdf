"""Grid forward (with the Jacobian) for the first k levels of the reference's grid, k = 1 .. 16, on ray-coherent points: the
differences are what each level costs (4096 rays x 64 samples, 2^19 table)."""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mipsfusion_amd import _lib, ops
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR

dev = torch.device("cuda:0")
PLS = float(2.0 ** (math.log2(16) / 15))
M = 4096 * 64
torch.manual_seed(0)
o = torch.rand(4096, 1, 3, device=dev) * 0.4 + 0.3
d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev), dim=-1) * 0.45
t = torch.sort(torch.rand(4096, 64, 1, device=dev), dim=1).values
x = (o + d * t).reshape(M, 3).clamp(0, 1).contiguous()


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for jac in (True, False):
    prev = 0.0
    print("with the Jacobian" if jac else "features only")
    for k in range(1, 17):
        meta = _lib.make_grid_meta(k, 2, 19, 16, PLS)
        params = (torch.rand(meta.n_params, device=dev) * 2 - 1) * 1e-2
        us = timeit(lambda: ops.hashgrid_fwd(x, params, meta, FEAT_LEVEL_MAJOR, with_jac=jac))
        size = meta.offsets[k] - meta.offsets[k - 1]
        print(f"levels 0..{k - 1:2d}: {us:6.1f} us  (+{us - prev:5.1f} for level {k - 1:2d}: res {meta.resolutions[k - 1]:4d}, {size:7d} entries)")
        prev = us
empty = timeit(lambda: ops.hashgrid_fwd(x[:64], params, meta, FEAT_LEVEL_MAJOR, with_jac=False))
print(f"64 samples (launch + events): {empty:.1f} us")
