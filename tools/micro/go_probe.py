import sys, os, time, types
sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mipsfusion_amd import ops, synth
from mipsfusion_amd.model import JointEncoding
from mipsfusion_amd.optim import FusedAdam
from mipsfusion_amd.helper_functions.geometry_helper import matrix_to_quaternion
from mipsfusion_amd.helper_functions.utils import get_loss_from_ret
dev = torch.device("cuda:0")
cfg = synth.config_headline()
bb = torch.from_numpy(np.array(cfg["mapping"]["bound"])); nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
model = JointEncoding(cfg, bb, nf).to(dev).train()
f = synth.make_frame(cfg, seed=1)
H, W = f["depth"].shape
rays = torch.cat([f["direction"], f["rgb"], f["depth"][..., None]], -1).reshape(-1, 7)
b = rays[torch.randperm(H * W)[:1000]].to(dev)
own = torch.zeros(1000, dtype=torch.int64, device=dev)
pose = f["c2w"].to(dev).float()
for prm in model.parameters(): prm.requires_grad_(False)
def go_loop(n, sleep=0.0):
    rot = torch.nn.Parameter(matrix_to_quaternion(pose[None, :3, :3])); trans = torch.nn.Parameter(pose[None, :3, 3].clone())
    popt = FusedAdam([{"params": rot, "lr": 1e-3}, {"params": trans, "lr": 1e-3}])
    if sleep: time.sleep(sleep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        ro, rd = ops.pose_rays(rot, trans, None, own, b[:, :3].contiguous())
        ret = model.forward(ro, rd, b[:, 3:6].contiguous(), b[:, 6:7].contiguous(), EMD_w=0., noise=torch.rand(1000, 64, device=dev))
        get_loss_from_ret(ret, cfg["training"]).backward()
        popt.step(zero_grad=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n
go_loop(20)
print("back-to-back ms/iter:", [round(go_loop(10), 3) for _ in range(4)])
print("after 50 ms host gap:", [round(go_loop(10, 0.05), 3) for _ in range(4)])
print("after 5 ms host gap:", [round(go_loop(10, 0.005), 3) for _ in range(4)])
