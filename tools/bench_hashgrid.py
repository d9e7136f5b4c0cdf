"""GPU microbench of the hash-grid kernels: ray-coherent vs random points, with/without dL/dx."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math, numpy as np, torch
from mipsfusion_amd import _lib, ops, synth
if os.environ.get("MIPSF_LIB_VARIANT"): _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libmipsf_%s.so" % os.environ["MIPSF_LIB_VARIANT"])
dev = torch.device("cuda:0")
PLS = float(2.0 ** (math.log2(16) / 15))
meta = _lib.make_grid_meta(16, 2, 19, 16, PLS)
M = 4096 * 64
params = ((torch.rand(meta.n_params, device=dev) * 2 - 1) * 1e-2)
torch.manual_seed(0)
# ray-coherent points: 4096 rays x 64 ascending samples through the unit cube
o = torch.rand(4096, 1, 3, device=dev) * 0.4 + 0.3
d = torch.nn.functional.normalize(torch.randn(4096, 1, 3, device=dev), dim=-1) * 0.45
t = torch.sort(torch.rand(4096, 64, 1, device=dev), dim=1).values
x_ray = (o + d * t).reshape(M, 3).clamp(0, 1).contiguous()
x_rnd = torch.rand(M, 3, device=dev)
dy = torch.randn(16, M, 2, device=dev) * 1e-3

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

only = sys.argv[1] if len(sys.argv) > 1 else None
for name, x in (("ray", x_ray), ("random", x_rnd)):
    if only and name != only: continue
    dp = torch.zeros_like(params); dx = torch.zeros(M, 3, device=dev)
    t_f = timeit(lambda: ops.hashgrid_fwd(x, params, meta, _lib.FEAT_LEVEL_MAJOR))
    t_b = timeit(lambda: ops.hashgrid_bwd(x, params, dy, dp, meta, _lib.FEAT_LEVEL_MAJOR, None))
    t_bx = timeit(lambda: ops.hashgrid_bwd(x, params, dy, dp, meta, _lib.FEAT_LEVEL_MAJOR, dx))
    print(f"{name:7s} fwd {t_f:8.1f} us ({1164*M/t_f/1e3:7.1f} GB/s)  bwd(no dx) {t_b:8.1f} us  bwd(dx) {t_bx:8.1f} us ({2188*M/t_bx/1e3:7.1f} GB/s)")
