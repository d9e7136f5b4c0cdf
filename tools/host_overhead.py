"""Where does host time go?  Runs the bench's mapping step at a tiny ray count (GPU work negligible)."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mipsfusion_amd import synth, ops
bench.N_RAYS = 512
cfg = synth.config_headline(); cfg["mapping"]["pixels_cur"] = 400
dev = torch.device("cuda:0")
model, frames, poses = bench.build_submap(cfg, dev, 0)
pool = bench.sample_pool(cfg, frames, 4)
loop = bench.MappingLoop(cfg, model, poses, pool, dev)
loop.noise = [torch.rand(bench.N_RAYS, 64, device=dev) for _ in pool]
for _ in range(20): loop.step()
torch.cuda.synchronize()
for prof in (False, True):
    ops.PROFILE = {} if prof else None
    t0 = time.perf_counter()
    for _ in range(100): loop.step()
    torch.cuda.synchronize()
    print(f"host-bound step (512 rays), event-profiling={prof}: {(time.perf_counter()-t0)*10:.3f} ms")
ops.PROFILE = None
pr = cProfile.Profile(); pr.enable()
for _ in range(50): loop.step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
