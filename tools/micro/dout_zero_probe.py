"""Share of (sample, level) pairs whose feature gradient is exactly zero on the benchmark's mapping iteration (those add
nothing to the grid gradient: candidates for being skipped by the scatter's routing)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from mipsfusion_amd import ops, synth

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
table, db, R = bench.build_ray_table(cfg, frames, dev)
idx_rows, idx_owner = bench.draw_index_sets(cfg, frames, db, R, 80)
loop = bench.MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev)
cap = {}
orig = ops.hashgrid_bwd


def grab(x, params, dout, dparams, meta, layout=ops.FEAT_AOS, *args, **kw):
    cap.update(dout=dout.clone(), layout=layout, M=x.shape[0], L=meta.n_levels)
    return orig(x, params, dout, dparams, meta, layout, *args, **kw)


ops.hashgrid_bwd = grab
for it in range(61):
    loop.step()
    if it in (0, 5, 20, 60):
        torch.cuda.synchronize()
        d = cap["dout"].view(cap["L"], cap["M"], 2) if cap["layout"] == ops.FEAT_LEVEL_MAJOR else cap["dout"].view(cap["M"], cap["L"], 2).transpose(0, 1)
        z = (d == 0).all(-1)                      # [L, M]
        print("iteration %2d: zero (sample, level) pairs %.3f; samples zero on every level %.3f; per level %s" % (
            it, z.float().mean().item(), z.all(0).float().mean().item(), " ".join("%.2f" % v for v in z.float().mean(1).tolist())))
        za = z.all(0)
        for tile in (16, 32, 64):
            print("    tiles of %2d consecutive samples that are zero throughout: %.3f" % (tile, za.view(-1, tile).all(1).float().mean().item()))
        per_ray = za.view(-1, 64)
        lead = (~per_ray).float().cumsum(1).eq(0).sum(1)          # zero samples before the first live one
        tail = (~per_ray).flip(1).float().cumsum(1).eq(0).sum(1)   # zero samples after the last live one
        print("    per ray: mean zero head %.1f, mean zero tail %.1f, rays entirely zero %.3f" % (
            lead.float().mean().item(), tail.float().mean().item(), per_ray.all(1).float().mean().item()))
