"""The tracking (pose-only) iteration of the graphed sequences, call by call: every C-ABI call of one iteration timed with an
event pair (eager), the sum, and the same iteration as one hipGraph replay."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
from mipsfusion_amd import _lib, synth  # noqa: E402
from mipsfusion_amd.graph import GraphedSteps, work_stream  # noqa: E402
from mipsfusion_amd.helper_functions.utils import get_loss_from_ret  # noqa: E402
from mipsfusion_amd.optim import FusedAdam  # noqa: E402
from replay import Recorder  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = work_stream(dev)
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
table, db, R = bench.build_ray_table(cfg, frames, dev)
idx_rows, idx_owner = bench.draw_index_sets(cfg, frames, db, R, 8)
loop = bench.MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev, capturable=True)
for _ in range(20):
    loop.step()
for prm in model.parameters():
    prm.requires_grad_(False)
ns = cfg["tracking"]["sample"]
rot = torch.nn.Parameter(loop.cur_rot.detach()[-1:].clone())
trans = torch.nn.Parameter(loop.cur_trans.detach()[-1:].clone())
popt = FusedAdam([{"params": rot, "lr": 1e-3}, {"params": trans, "lr": 1e-3}], capturable=True)
noise = loop.noise[0][:ns]
own = torch.zeros(ns, dtype=torch.int64, device=dev)
rows_go = loop.idx_rows[0][:ns].contiguous()


def go():
    ret = model.forward_from_table(loop.table, rows_go, rot, trans, None, own, noise, EMD_w=0., accumulate_in_place=True)
    get_loss_from_ret(ret, cfg["training"]).backward()
    popt.step(zero_grad=True)


with torch.cuda.stream(stream):
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    base = _lib.lib()
    rec = Recorder(base)
    _lib._lib = rec
    go()
    torch.cuda.synchronize()
    _lib._lib = base
    calls = rec.calls
    reps = 30
    evs = [[] for _ in calls]
    for r in range(reps + 3):
        for k, (n, a) in enumerate(calls):
            if r >= 3:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            getattr(base, n)(*a)
            if r >= 3:
                e1.record()
                evs[k].append((e0, e1))
    torch.cuda.synchronize()
    tot = 0.0
    print(f"tracking iteration: {ns} rays x {noise.shape[1]} samples = {ns * noise.shape[1]} samples, map frozen")
    for k, (n, _a) in enumerate(calls):
        us = sum(a.elapsed_time(b) for a, b in evs[k]) / len(evs[k]) * 1e3
        tot += us
        print(f"  {k:2d} {n[6:]:32s} {us:7.1f} us")
    print(f"  sum of the calls {tot:.1f} us")
g = GraphedSteps(lambda k: go(), 1, stream=stream)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(50):
    g.replay()
torch.cuda.synchronize()
print(f"  one hipGraph replay {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
