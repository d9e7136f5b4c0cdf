"""Times mipsf_hashgrid_route ALONE on the benchmark's sample positions (ablation builds of the routing kernel leave
garbage records behind: nothing downstream may run on them).  usage (GPU box): python tools/micro/route_probe.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import numpy as np
from mipsfusion_amd import _lib, ops, synth
from mipsfusion_amd.model import JointEncoding

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = synth.config_headline()
bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
model = JointEncoding(cfg, bb, nf).to(dev)
meta = model.embed_fn.meta
# 4096 rays x 64 consecutive samples through the unit box (the benchmark's shape; no kernel of the library runs here)
torch.manual_seed(0)
R, S = 4096, 64
o = torch.rand(R, 1, 3, device=dev) * 0.2 + 0.4
d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
tt = torch.linspace(0.0, 0.35, S, device=dev).view(1, S, 1)
x = (o + d * tt).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
M = x.shape[0]
lib = _lib.lib()
n = _lib.buffer_size(_lib.SIZE_HASHGRID_BWD_SCRATCH, M, 0, 0, meta)
scratch = torch.zeros(n, dtype=torch.float32, device=dev)
for _ in range(5):
    ops.check(lib.mipsf_hashgrid_route(ops.dptr(x), ops.dptr(scratch), M, C.byref(meta), ops.stream_ptr()), "route")
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    ops.check(lib.mipsf_hashgrid_route(ops.dptr(x), ops.dptr(scratch), M, C.byref(meta), ops.stream_ptr()), "route")
b.record()
torch.cuda.synchronize()
print("route (zero + route + scan) %.1f us per call, M = %d" % (a.elapsed_time(b) * 1e3 / 50, M))
