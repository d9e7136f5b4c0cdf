"""Per-kernel times of a RandomOptimizer frame (5 rounds, eager, ops.PROFILE events), per decoder arithmetic.
    python tools/micro/ro_round_times.py"""
import os
import sys
import types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from mipsfusion_amd import ops, synth
from mipsfusion_amd.RandomOptimizer import RandomOptimizer

dev = torch.device("cuda:0")
cfg = synth.config_headline()
cfg["tracking"]["RO"].setdefault("initial_scaling_factor", 0.02)
cfg["tracking"]["RO"].setdefault("rescaling_factor", 0.5)
cfg["tracking"].setdefault("ignore_edge_W", 20)
cfg["tracking"].setdefault("ignore_edge_H", 20)
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
model.eval()
H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
f = synth.make_frame(cfg, seed=1)
ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=f["direction"])
np.random.seed(0)
ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
init = f["c2w"].clone()
for prec in ("bf16x6", "f16x3", "f16"):
    ro.decoder_precision = prec
    for _ in range(3):
        ro.optimize(model, f["depth"], init, None, n_iter=5)
    torch.cuda.synchronize()
    ops.PROFILE = {}
    for _ in range(5):
        ro.optimize(model, f["depth"], init, None, n_iter=5)
    torch.cuda.synchronize()
    s = ops.profile_summary()
    ops.PROFILE = None
    tot = sum(n * ms for n, ms in s.values()) / 25
    print(f"{prec}: {tot * 1e3:.1f} us of kernels per round: " + ", ".join(f"{k} {ms * 1e3:.1f}" + (f" x{n // 25}" if n != 25 else "") for k, (n, ms) in sorted(s.items(), key=lambda kv: -kv[1][1])))
