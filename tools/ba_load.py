"""A second process that keeps ONE GPU busy with local-BA mapping steps (the role of the reference's InactiveMap process,
mipsfusion.py:661-667, InactiveMap.py:203-308): prints READY once the steps run, then iterates until it is terminated or
--seconds elapse.  Used by tests/test_gpu_configs.py::test_mapping_and_tracking_beside_a_second_process_on_the_same_gpu."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mipsfusion_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--only", default=None, help="regex: replay only the matching C-ABI calls of one recorded step (diagnosis: which "
                                               "kind of neighbour kernel matters); the optimiser calls are always left out")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=5)
table, db, R = bench.build_ray_table(cfg, frames, dev)
rows, owner = bench.draw_index_sets(cfg, frames, db, R, 32)
loop = bench.MappingLoop(cfg, model, poses, table, rows, owner, dev)
for _ in range(5):
    loop.step()
torch.cuda.synchronize()
if args.only:
    import re
    from mipsfusion_amd import _lib
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from replay import Recorder
    base = _lib.lib()
    rec = Recorder(base)
    _lib._lib = rec
    loop.i = 4
    loop.step()
    torch.cuda.synchronize()
    _lib._lib = base
    calls = [(n_, a) for n_, a in rec.calls if "adam" not in n_ and re.search(args.only, n_)]
    print("READY", flush=True)
    t0, n = time.time(), 0
    while time.time() - t0 < args.seconds:
        for _ in range(20):
            for n_, a in calls:
                assert getattr(base, n_)(*a) == 0, n_
        torch.cuda.synchronize()
        n += 20
    print(f"ba_load: {n} x {sorted({c[0][6:] for c in calls})} in {time.time() - t0:.1f} s", flush=True)
    sys.exit(0)
print("READY", flush=True)
t0, n = time.time(), 0
while time.time() - t0 < args.seconds:
    for _ in range(20):
        loop.step()
    torch.cuda.synchronize()
    n += 20
print(f"ba_load: {n} mapping steps in {time.time() - t0:.1f} s", flush=True)
