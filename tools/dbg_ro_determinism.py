"""Are the SDF-only decoder forward and a RandomOptimizer frame bit-reproducible run to run?  (the particle-split check of
bench.py --gpus 2 compares poses with torch.equal)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mipsfusion_amd import ops, synth
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR
from mipsfusion_amd.RandomOptimizer import RandomOptimizer
dev = torch.device("cuda:0")
cfg = synth.config_headline()
cfg["tracking"]["RO"].setdefault("initial_scaling_factor", 0.02)
cfg["tracking"]["RO"].setdefault("rescaling_factor", 0.5)
cfg["tracking"].setdefault("ignore_edge_W", 20); cfg["tracking"].setdefault("ignore_edge_H", 20)
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
model.eval()
packed16 = ops.decoder_pack16(model.decoder.ordered_parameters())
for M in (768000, 384000, 64000, 1000):
    x = torch.rand(M, 3, device=dev)
    feat = torch.randn(16, M, 2, device=dev) * 1e-2
    ref = ops.decoder_fwd_sdf(None, feat, FEAT_LEVEL_MAJOR, x, None, M, precision="f16x3", packed16=packed16).clone()
    full = ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save=False, precision="f16x3", packed16=packed16)[0].clone()
    bad = badf = 0
    for r in range(60):
        o = ops.decoder_fwd_sdf(None, feat, FEAT_LEVEL_MAJOR, x, None, M, precision="f16x3", packed16=packed16)
        bad += int(not torch.equal(o, ref))
        o2 = ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save=False, precision="f16x3", packed16=packed16)[0]
        badf += int(not torch.equal(o2, full))
        if not torch.equal(o, ref) and bad <= 2:
            d = (o - ref).abs()
            i = int(d.argmax()); print("   sdf differs at sample", i, "tile", i // 32, "lane", i % 32, float(d.max()), "n differing", int((d > 0).sum()))
    sub = ops.decoder_fwd_sdf(None, feat[:, :M // 2].contiguous(), FEAT_LEVEL_MAJOR, x[:M // 2].contiguous(), None, M // 2, precision="f16x3", packed16=packed16)
    print(f"M={M}: sdf-only runs differing from the first: {bad}/60, full forward: {badf}/60; first half alone equals the first half of the batch: {bool(torch.equal(sub, ref[:M // 2]))}")
H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
f = synth.make_frame(cfg, seed=1)
ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=f["direction"])
np.random.seed(0)
ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
init = f["c2w"].clone(); init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
p0 = ro.optimize(model, f["depth"], init, None, n_iter=5).clone()
bad = sum(int(not torch.equal(ro.optimize(model, f["depth"], init, None, n_iter=5), p0)) for _ in range(40))
print(f"RandomOptimizer frame (5 rounds, f16x3): {bad}/40 runs differ from the first")
