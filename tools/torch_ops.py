"""Which torch (non-mipsf) kernels run inside one mapping step?  torch.profiler over 10 eager steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from mipsfusion_amd import synth
bench.N_RAYS = 512
cfg = synth.config_headline(); cfg["mapping"]["pixels_cur"] = 400
dev = torch.device("cuda:0")
model, frames, poses = bench.build_submap(cfg, dev, 0)
pool = bench.sample_pool(cfg, frames, 4)
loop = bench.MappingLoop(cfg, model, poses, pool, dev)
loop.noise = [torch.rand(bench.N_RAYS, 64, device=dev) for _ in pool]
for _ in range(20): loop.step()
torch.cuda.synchronize()
N = 10
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N): loop.step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    if e.key.startswith("aten::") and e.device_time_total > 0 and any(k in e.key for k in ("fill", "zero", "add", "mul", "copy", "div", "sub", "neg", "sum", "cat", "index", "clone")):
        rows.append((e.count / N, e.key, [s for s in e.stack][:6]))
rows.sort(key=lambda r: -r[0])
for c, k, st in rows[:40]:
    print(f"{c:5.1f}/step {k:24s} {' <- '.join(s.split('/')[-1][:70] for s in st)}")
