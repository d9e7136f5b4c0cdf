"""cProfile of the unchanged caller's iteration (bench.UnchangedCallerLoop, indices pre-drawn): where the HOST time of forward /
backward / optimiser goes.  python tools/prof_unchanged.py"""
import cProfile
import os
import pstats
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench            # FIRST: it confines the process to one NUMA node and sizes OpenMP before torch is imported
import torch
from mipsfusion_amd import synth

dev = torch.device("cuda:0")
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
table, db, R = bench.build_ray_table(cfg, frames, dev)
loop = bench.UnchangedCallerLoop(cfg, model, frames, poses, table, db, R, dev)
rows, owner = bench.draw_index_sets(cfg, frames, db, R, 8)
for k in range(5):
    loop.iterate((rows[k % 8], owner[k % 8]))
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for k in range(40):
    loop.iterate((rows[k % 8], owner[k % 8]))
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
# the autograd engine runs backward on its own thread: profile it there as well
import threading
prof2 = cProfile.Profile()
orig = torch.autograd.Function.backward
from mipsfusion_amd.model import scene_rep
for cls in (scene_rep._QueryFn, scene_rep._RenderFn, scene_rep._PlaceFn):
    f = cls.backward
    def wrap(ctx, *a, _f=f):
        prof2.enable()
        try:
            return _f(ctx, *a)
        finally:
            prof2.disable()
    cls.backward = staticmethod(wrap)
for k in range(40):
    loop.iterate((rows[k % 8], owner[k % 8]))
torch.cuda.synchronize()
print("=========== inside the autograd Functions' backward (engine thread), 40 iterations")
pstats.Stats(prof2).sort_stats("cumulative").print_stats(35)

# torch.profiler over 20 iterations: the operators and kernels of the backward pass by name (CPU time on the engine thread,
# device time), to see what the 1.1 ms of run_backward is made of
if os.environ.get("TORCH_PROF", "1") == "1":
    from torch.profiler import profile, ProfilerActivity
    for cls in (scene_rep._QueryFn, scene_rep._RenderFn, scene_rep._PlaceFn):
        pass
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for k in range(20):
            loop.iterate((rows[k % 8], owner[k % 8]))
        torch.cuda.synchronize()
    print("=========== torch.profiler, 20 iterations, by CPU time")
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40, max_name_column_width=60))
    print("=========== by device time")
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60))
