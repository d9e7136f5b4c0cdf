import sys, os, time, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from mipsfusion_amd import synth
from mipsfusion_amd.model import JointEncoding
from mipsfusion_amd.RandomOptimizer import RandomOptimizer
dev = torch.device("cuda:0")
cfg = synth.config_headline()
cfg["tracking"]["RO"].setdefault("initial_scaling_factor", 0.02)
cfg["tracking"]["RO"].setdefault("rescaling_factor", 0.5)
cfg["tracking"].setdefault("ignore_edge_W", 20); cfg["tracking"].setdefault("ignore_edge_H", 20)
bb = torch.from_numpy(np.array(cfg["mapping"]["bound"])); nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
model = JointEncoding(cfg, bb, nf).to(dev).eval()
f = synth.make_frame(cfg, seed=1)
H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=f["direction"])
init = f["c2w"].to(dev).float()
for prec in ("f16x3", "f16", "f16x3", "f16"):
    ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
    ro.decoder_precision = prec
    for _ in range(3):
        ro.optimize(model, f["depth"], init, None, n_iter=5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ro.optimize(model, f["depth"], init, None, n_iter=5)
    torch.cuda.synchronize()
    print(os.path.basename(os.environ.get("MIPSF_LIB", "base")), prec, "ms per RO round:", round((time.perf_counter() - t0) / 50 * 1e3, 4))
