"""Hash-grid backward on the headline workload's sample distribution: records per (sample, level), per-level timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mipsfusion_amd import ops, synth, _lib
from mipsfusion_amd.model import JointEncoding
dev = torch.device("cuda:0")
cfg = synth.config_headline()
bb = torch.from_numpy(np.array(cfg["mapping"]["bound"])); nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
model = JointEncoding(cfg, bb, nf).to(dev).train()
meta = model.embed_fn.meta
L = meta.n_levels
offs = [meta.offsets[l] for l in range(L + 1)]
res = [meta.resolutions[l] for l in range(L)]
# rays like the bench's: 4096 rays x 64 samples, consecutive samples along a ray
torch.manual_seed(0)
R, S = 4096, 64
o = torch.rand(R, 1, 3, device=dev) * 0.2 + 0.4
d = torch.nn.functional.normalize(torch.randn(R, 1, 3, device=dev), dim=-1)
t = torch.linspace(0.0, 0.35, S, device=dev).view(1, S, 1)
x = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
M = x.shape[0]
idx = ops.hashgrid_indices(x, meta).long()
print("level res size slices rec/sample  distinct-cells/ray")
for l in range(L):
    size = offs[l + 1] - offs[l]
    dense = (res[l] + 1) ** 3 <= size + 8   # heuristic print only
    one = size <= 10240
    sl = idx[:, l, :] >> (31 if one else 13)
    srt = sl.sort(dim=1).values
    distinct = 1 + (srt[:, 1:] != srt[:, :-1]).sum(1)
    cells = idx[:, l, 0].view(R, S)
    runs = 1 + (cells[:, 1:] != cells[:, :-1]).sum(1)
    print(f"{l:2d} {res[l]:5d} {size:8d} {1 if one else (size + 8191) // 8192:4d} {distinct.float().mean().item():6.2f}   {runs.float().mean().item():6.1f}")
dout = torch.randn(L, M, 2, device=dev) * 1e-4
dparams = torch.zeros_like(model.embed_fn.params)
for _ in range(12):
    ops.hashgrid_bwd(x, model.embed_fn.params.detach(), dout, dparams, meta, _lib.FEAT_LEVEL_MAJOR)
torch.cuda.synchronize()
