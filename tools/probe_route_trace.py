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


def grab(x, params, dout, dparams, meta, layout=ops.FEAT_AOS, *args, **kw):
    cap.update(x=x.clone(), dout=dout.clone(), meta=meta, layout=layout, params=params)
    return orig(x, params, dout, dparams, meta, layout, *args, **kw)


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
n = _lib.buffer_size(_lib.SIZE_HASHGRID_BWD_SCRATCH, M, 0, 0, meta)
scratch = torch.zeros(n, dtype=torch.float32, device=dev)
dparams = torch.zeros_like(params)
ARGS = _lib.HashgridBwdArgs.new(M=M, x=ops.dptr(x), params=ops.dptr(params), dout=ops.dptr(dout), dparams=ops.dptr(dparams),
                                scratch=ops.dptr(scratch), meta=C.pointer(meta), feat_layout=layout)
for rep in range(3):
    ops.check(lib.mipsf_hashgrid_bwd(C.byref(ARGS), ops.stream_ptr()), "bwd")
torch.cuda.synchronize()
rows = scratch[off:off + 20 * n_rows.value].view(torch.int64).cpu().numpy().reshape(-1, 10)
names = ["liveness loads + ballots", "scan + compaction (3 barriers)", "x gather, locate, group, rank", "barrier",
         "bin atomics issued + prefix", "barrier (atomics return)", "records -> stage", "barrier", "write-out"]
grp, nlv = rows[:, 9] & 0xff, np.maximum(rows[:, 9] >> 8, 1)
print(f"{rows.shape[0]} waves, {int(nlv.max())} levels per workgroup; cycles per phase (mean over all waves; the first two phases are paid "
      f"once per workgroup, the others are sums over its levels)")
r = rows[:, :9].astype(np.float64)
print(f"  total {r.sum(1).mean():7.0f} per workgroup-wave = {(r.sum(1) / nlv).mean():7.0f} per level: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, r.mean(0))))
print("  per level group: " + " ".join(f"{g_}:{rows[grp == g_, :9].sum(1).mean():.0f}" for g_ in sorted(set(grp.tolist()))))
