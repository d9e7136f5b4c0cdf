"""GPU diagnostic: locate where the GPU gradient of a headline-config subset departs from the oracle."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mipsfusion_amd import synth, ops, _lib
from mipsfusion_amd.model import JointEncoding
from oracle import path_cpu

dev = torch.device("cuda:0")
hash_size = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = synth.config_headline(); cfg["grid"]["hash_size"] = hash_size
bb = torch.from_numpy(np.array(cfg["mapping"]["bound"])); nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
torch.manual_seed(0)
m = JointEncoding(cfg, bb, nf).to(dev).train()
with torch.no_grad():
    m.embed_fn.params.copy_((torch.randn(m.embed_fn.params.shape) * 0.2).to(dev))
frame = synth.make_frame(cfg, seed=0); H, W = frame["depth"].shape
random.seed(0); idx = torch.tensor(random.sample(range(H * W), 192))
ro, rd, rgb, d = synth.ray_batch(frame, idx, frame["c2w"])
noise = torch.rand(192, 64)
cpu = path_cpu.CpuScene(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"])
cpu.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})

def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))

# oracle with intermediate grads
o = cpu.train_forward(ro, rd, rgb, d, noise, 0.01)
o["raw"].retain_grad()
lo = path_cpu.total_loss(o, cfg["training"]); lo.backward()
# gpu
ret = m.forward(ro.to(dev), rd.to(dev), rgb.to(dev), d.to(dev), noise=noise.to(dev))
lg = path_cpu.total_loss(ret, cfg["training"]); lg.backward()
print("loss", float(lg), float(lo), "rel", abs(float(lg) - float(lo)) / abs(float(lo)))
print("grid grad rel err", rel(m.embed_fn.params.grad, cpu.embed_fn.params.grad))
for k, v in m.decoder.named_parameters():
    print("  dec", k, rel(v.grad, dict(cpu.decoder.named_parameters())[k].grad))
# stage-wise: render backward on oracle raw
rc = m._rc(43, 21, 0.01)
N, S = 192, 64
tables = m._linspace_tables(dev, True)
z, xn, counts = ops.sample_rays(ro.to(dev), rd.to(dev), d.to(dev), noise.to(dev), tables, rc, N, S)
print("z equal", torch.equal(z.cpu(), o["z_vals"]))
raw_cpu = o["raw"].detach().reshape(N * S, 10).contiguous().to(dev)
_, _, _, _, _, _, losses = ops.render_fwd(raw_cpu, z, rgb.to(dev), d.to(dev), counts, rc, N, S, True)
tr = cfg["training"]
gl = torch.tensor([tr["rgb_weight"], tr["depth_weight"], tr["sdf_weight"], tr["fs_weight"], 0, 0, 0, 0], dtype=torch.float32, device=dev)
draw = ops.render_bwd(raw_cpu, z, rgb.to(dev), d.to(dev), counts, losses, rc, gl, None, None, N, S)
dref = o["raw"].grad.reshape(N * S, 10)
print("render_bwd(draw) on oracle raw: rel err", rel(draw, dref))
for c in range(10):
    print("   col", c, rel(draw[:, c], dref[:, c]), float(dref[:, c].abs().max()))
bad = ((draw.cpu() - dref).abs().max(1).values > 1e-3 * dref.abs().max()).nonzero().squeeze(-1)
print("   samples off by >1e-3*max:", bad.numel(), bad[:10].tolist())
# grid grad by level
offs = list(m.embed_fn.meta.offsets[:17])
gg, gc = m.embed_fn.params.grad.cpu(), cpu.embed_fn.params.grad
for l in range(16):
    a, b = gg[2 * offs[l]:2 * offs[l + 1]], gc[2 * offs[l]:2 * offs[l + 1]]
    print("  level", l, "err/max_level", float((a - b).abs().max() / b.abs().max()), "max", float(b.abs().max()), "nnz", int((b != 0).sum()))
