import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_parity import make_scene, cfg_for, T
from tests.conftest import load_golden
dev = torch.device("cuda:0")
g = load_golden("scene_s75.npz"); cfg = cfg_for("scene_s75.npz")
for prec in ("f32", "f16x3"):
    m = make_scene(g, cfg, dev).train(); m.decoder_precision = prec
    ro = T(g["rays_o"]).to(dev).requires_grad_(True); rd = T(g["rays_d"]).to(dev).requires_grad_(True)
    ret = m.forward(ro, rd, T(g["target_rgb"]).to(dev), T(g["target_d"]).to(dev), EMD_w=0.01, noise=T(g["noise"]).to(dev))
    tr = cfg["training"]
    loss = tr["rgb_weight"] * ret["rgb_loss"] + tr["sdf_weight"] * ret["sdf_loss"] + tr["fs_weight"] * ret["fs_loss"]
    loss.backward()
    d = (ro.grad.cpu().numpy() - g["emd.d_rays_o"]); scale = np.abs(g["emd.d_rays_o"]).max()
    per_ray = np.abs(d).max(1) / scale
    print(prec, "rays off by > 5e-4 of max:", np.nonzero(per_ray > 5e-4)[0].tolist(), "worst", per_ray.max(), "median", np.median(per_ray))
    dd = (rd.grad.cpu().numpy() - g["emd.d_rays_d"]); print("   d rays_d worst", np.abs(dd).max() / np.abs(g["emd.d_rays_d"]).max())
    raw_ref = g["eval.raw"] if "eval.raw" in g.files else None
