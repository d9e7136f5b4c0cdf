"""RandomOptimizer.optimize run to run, many times (run two instances concurrently to share the GPU): which stage differs first?"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from mipsfusion_amd import ops, synth
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR
from mipsfusion_amd.RandomOptimizer import RandomOptimizer, _POINT_MAJOR
dev = torch.device("cuda:0")
cfg = synth.config_headline()
cfg["tracking"]["RO"].setdefault("initial_scaling_factor", 0.02)
cfg["tracking"]["RO"].setdefault("rescaling_factor", 0.5)
cfg["tracking"].setdefault("ignore_edge_W", 20); cfg["tracking"].setdefault("ignore_edge_H", 20)
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
model.eval()
H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
f = synth.make_frame(cfg, seed=1)
ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=f["direction"])
np.random.seed(0)
ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
init = f["c2w"].clone(); init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
p0 = ro.optimize(model, f["depth"], init, None, n_iter=5).clone()
bad = 0
for k in range(N):
    p = ro.optimize(model, f["depth"], init, None, n_iter=5)
    if not torch.equal(p, p0):
        bad += 1
        if bad <= 3:
            print(f"  run {k}: pose differs by {float((p - p0).abs().max()):.3e}")
print(f"optimize(): {bad}/{N} runs differ from the first")
# stage by stage on fixed inputs: one round's kernels, each output compared with its first value
ws = model.decoder.ordered_parameters()
packed = ops.decoder_pack16(ws)
rows, cols = ro.row_indices, ro.col_indices
td = f["depth"][rows, cols].to(dev, torch.float32).contiguous()
state = torch.zeros(ops.RO_STATE_FLOATS, device=dev)
state[0:9], state[9:12], state[12:18] = init[:3, :3].reshape(9).to(dev), init[:3, 3].to(dev), 0.02
rc = model._rc(1, 0)
first = None
cnt = {"xn": 0, "feat": 0, "sdf": 0, "fit": 0, "packed": 0}
for k in range(N):
    pk = ops.decoder_pack16(ws)
    xn, pst7 = ops.ro_particles(ro.pre_sampled_particle, state.clone(), ro._dirs[0], td, rc, point_major=_POINT_MAJOR)
    feat = ops.hashgrid_fwd(xn, model.embed_fn.params.detach(), model.embed_fn.meta, FEAT_LEVEL_MAJOR)
    sdf = ops.decoder_fwd_sdf(None, feat, FEAT_LEVEL_MAJOR, xn, None, xn.shape[0], precision="f16x3", packed16=pk)
    fit = ops.ro_fitness(sdf.view(ro.particle_size, -1, 1), td, ro.trunc_value, point_major=_POINT_MAJOR)
    cur = {"xn": xn, "feat": feat, "sdf": sdf, "fit": fit, "packed": pk}
    if first is None:
        first = {a: b.clone() for a, b in cur.items()}
    else:
        for a in cnt:
            if not torch.equal(cur[a], first[a]):
                cnt[a] += 1
                if cnt[a] <= 2:
                    d = (cur[a].float() - first[a].float()).abs().reshape(-1)
                    nz = (d > 0).nonzero().reshape(-1)
                    print(f"  iteration {k}: {a} differs: {int((d > 0).sum())} elements, max {float(d.max()):.3e}, first at {int(nz[0])}, last at {int(nz[-1])}")
                    if a == "xn":
                        smp = torch.unique(nz // 3)
                        P_ = ro.particle_size
                        print("     points", torch.unique(smp // P_).tolist()[:40], "particles", torch.unique(smp % P_).tolist()[:40])
                        print("     values now", cur[a].reshape(-1)[nz[:6]].tolist(), "first", first[a].reshape(-1)[nz[:6]].tolist())
print("stage outputs differing from their first value:", cnt, "of", N)
