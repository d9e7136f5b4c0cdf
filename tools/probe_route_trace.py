"""Phase timeline of the scatter's ROUTING kernel on the headline step's own (x, dL/dy): needs a library built with
-DMIPSF_RT_TRACE (tools/micro/variant.sh rtrace hashgrid -DMIPSF_RT_TRACE; MIPSF_LIB=tools/micro/libv_rtrace.so).
Prints cycles per phase of a workgroup's waves, per level class."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from mipsfusion_amd import _lib, ops, synth

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
table, db, R = bench.build_ray_table(cfg, frames, dev)
idx_rows, idx_owner = bench.draw_index_sets(cfg, frames, db, R, 64)
loop = bench.MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev)
for _ in range(55):
    loop.step()
cap = {}
orig = ops.hashgrid_bwd


def grab(x, params, dout, dparams, meta, layout=ops.FEAT_AOS, dx=None, routed=None, dparams_zero=False):
    cap.update(x=x.clone(), dout=dout.clone(), meta=meta, layout=layout, params=params)
    return orig(x, params, dout, dparams, meta, layout, dx, routed, dparams_zero)


ops.hashgrid_bwd = grab
loop.step()
torch.cuda.synchronize()
ops.hashgrid_bwd = orig
x, dout, meta, layout, params = cap["x"], cap["dout"], cap["meta"], cap["layout"], cap["params"]
M = x.shape[0]
lib = _lib.lib()
lib.mipsf_hashgrid_rt_trace_words.restype = C.c_uint64
lib.mipsf_hashgrid_rt_trace_words.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
n_rows = C.c_uint32()
off = lib.mipsf_hashgrid_rt_trace_words(C.byref(meta), M, C.byref(n_rows))
n = lib.mipsf_hashgrid_bwd_scratch_floats(C.byref(meta), M, 0)
scratch = torch.zeros(n, dtype=torch.float32, device=dev)
dparams = torch.zeros_like(params)
for rep in range(3):
    ops.check(lib.mipsf_hashgrid_bwd(ops.dptr(x), ops.dptr(params), ops.dptr(dout), ops.dptr(dparams), None,
                                     ops.dptr(scratch), M, C.byref(meta), layout, ops.stream_ptr()), "bwd")
torch.cuda.synchronize()
rows = scratch[off:off + 20 * n_rows.value].view(torch.int64).cpu().numpy().reshape(-1, 10)
names = ["liveness loads + ballots", "scan + compaction (3 barriers)", "x gather, locate, group, rank", "barrier",
         "bin atomics issued + prefix", "barrier (atomics return)", "records -> stage", "barrier", "write-out"]
lvl = rows[:, 9]
L = int(lvl.max()) + 1
print(f"{rows.shape[0]} waves; cycles per phase (mean over the waves of a level class)")
for label, sel in (("dense levels", lvl < 9), ("hashed levels", lvl >= 9), ("all", lvl >= 0)):
    r = rows[sel, :9].astype(np.float64)
    tot = r.sum(1).mean()
    print(f"  {label:14s} total {tot:7.0f}: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, r.mean(0))))
print("  per level totals: " + " ".join(f"{l}:{rows[lvl == l, :9].sum(1).mean():.0f}" for l in range(L)))
