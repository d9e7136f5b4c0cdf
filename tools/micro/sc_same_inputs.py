"""Scatter on IDENTICAL seeded inputs (ray-coherent points, a gradient whose tails are zero), for comparing library builds from
separate processes: dumps the gradient of levels 9..10 and a checksum of every level.  MIPSF_LIB selects the build.
    SC_OUT=gpurun_out/x.pt python tools/micro/sc_same_inputs.py"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mipsfusion_amd import _lib, ops
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR

dev = torch.device("cuda:0")
PLS = float(2.0 ** (math.log2(16) / 15))
N, S = 4096, 64
M = N * S
g = torch.Generator().manual_seed(3)
o = torch.rand(N, 1, 3, generator=g) * 0.4 + 0.3
d = torch.nn.functional.normalize(torch.randn(N, 1, 3, generator=g), dim=-1) * 0.45
t = torch.sort(torch.rand(N, S, 1, generator=g), dim=1).values
x = (o + d * t).reshape(M, 3).clamp(0, 1).contiguous().to(dev)
dout = (torch.randn(16, M, 2, generator=g) * 1e-6)
live = (torch.arange(S)[None, :] < torch.randint(20, 64, (N, 1), generator=g)).reshape(M)       # ray tails dead
dout = (dout * live[None, :, None]).contiguous().to(dev)
meta = _lib.make_grid_meta(16, 2, 19, 16, PLS)
params = torch.zeros(meta.n_params, device=dev)
outs = []
for rep in range(2):
    gp = torch.zeros(meta.n_params, device=dev)
    ops.hashgrid_bwd(x, params, dout, gp, meta, FEAT_LEVEL_MAJOR)
    torch.cuda.synchronize()
    outs.append(gp)
offs = [int(v) * 2 for v in meta.offsets[:17]]
print("lib", os.environ.get("MIPSF_LIB", "in-tree"), "run-to-run differing entries by level:",
      [int((outs[0][offs[l]:offs[l + 1]] != outs[1][offs[l]:offs[l + 1]]).sum()) for l in range(16)])
print("nonzero by level:", [int((outs[0][offs[l]:offs[l + 1]] != 0).sum()) for l in range(16)])
print("fp64 sums by level:", [f"{float(outs[0][offs[l]:offs[l + 1]].double().sum()):.12e}" for l in range(16)])
if os.environ.get("SC_OUT"):
    os.makedirs(os.path.dirname(os.environ["SC_OUT"]), exist_ok=True)
    torch.save(outs[0][offs[9]:offs[11]].cpu(), os.environ["SC_OUT"])
