"""Timeline of the hash-grid scatter's work items on the benchmark's own (x, dL/dy): needs a library built with
-DMIPSF_SC_TRACE (GPU box: `cd mipsfusion_amd/csrc && touch hashgrid.hip && make EXTRA=-DMIPSF_SC_TRACE`).
Prints, per level: items, records, time per item, and the kernel's per-XCD / per-CU occupancy over time."""
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
idx_rows, idx_owner = bench.draw_index_sets(cfg, frames, db, R, 40)
loop = bench.MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev)
for _ in range(30):
    loop.step()
cap = {}
orig = ops.hashgrid_bwd


def grab(x, params, dout, dparams, meta, layout=ops.FEAT_AOS, dx=None, routed=None):
    cap.update(x=x.clone(), dout=dout.clone(), meta=meta, layout=layout, params=params)
    return orig(x, params, dout, dparams, meta, layout, dx, routed)


ops.hashgrid_bwd = grab
import mipsfusion_amd.model.scene_rep as sr
for mod in (sr,):
    if hasattr(mod, "ops"):
        mod.ops.hashgrid_bwd = grab
loop.step()
torch.cuda.synchronize()
assert cap, "hashgrid_bwd was not called through ops"
x, dout, meta, layout, params = cap["x"], cap["dout"], cap["meta"], cap["layout"], cap["params"]
M = x.shape[0]
lib = _lib.lib()
lib.mipsf_hashgrid_trace_words.restype = C.c_uint64
lib.mipsf_hashgrid_trace_words.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
n_rows = C.c_uint32()
bin0 = (C.c_uint32 * 33)()
off = lib.mipsf_hashgrid_trace_words(C.byref(meta), M, C.byref(n_rows), bin0)
n = lib.mipsf_hashgrid_bwd_scratch_floats(C.byref(meta), M, 0)
scratch = torch.zeros(n, dtype=torch.float32, device=dev)
dparams = torch.zeros_like(params)
L = meta.n_levels
for rep in range(3):
    scratch[off:off + 16 * n_rows.value].zero_()
    ops.check(lib.mipsf_hashgrid_route(ops.dptr(x), ops.dptr(scratch), M, C.byref(meta), ops.stream_ptr()), "route")
    ops.check(lib.mipsf_hashgrid_bwd_routed(ops.dptr(x), ops.dptr(params), ops.dptr(dout), ops.dptr(dparams), None,
                                            ops.dptr(scratch), M, C.byref(meta), layout, ops.stream_ptr()), "bwd")
torch.cuda.synchronize()
tr = scratch[off:off + 16 * n_rows.value].view(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 16)
tr = tr[tr[:, 4] != 0]
item, nrec = tr[:, 0], tr[:, 1]
t0 = tr[:, 2].astype(np.uint64) | (tr[:, 3].astype(np.uint64) << 32)
t1 = tr[:, 4].astype(np.uint64) | (tr[:, 5].astype(np.uint64) << 32)
xcc, hw = tr[:, 6] & 0xf, tr[:, 7]
base = t0.min()
b = ((t0 - base) * 0.01).astype(np.float64)      # us (100 MHz)
e = ((t1 - base) * 0.01).astype(np.float64)
bins = item & 0xffff
b0 = np.array([bin0[l] for l in range(L + 1)])
level = np.searchsorted(b0, bins, side="right") - 1
print("M", M, "items", len(tr), "kernel span %.1f us" % e.max())
print("level items  records   mean_us  max_us   first_start last_end   rec/us/item   | wave 0: zero, loop end, sync, iterations (us from item start)")
for l in range(L):
    m = level == l
    if not m.any():
        continue
    d = e[m] - b[m]
    ph = tr[m][:, 8:16].astype(np.float64).mean(0) * 0.01
    print("%3d %6d %9d %8.1f %7.1f %10.1f %9.1f %10.0f   | %5.1f %5.1f %5.1f  it: %s" % (
        l, m.sum(), nrec[m].sum(), d.mean(), d.max(), b[m].min(), e[m].max(), (nrec[m] / np.maximum(d, 0.01)).mean(),
        ph[0], ph[1], ph[2], " ".join("%.1f" % v for v in ph[3:])))
cu = (hw >> 8) & 0xf | ((hw >> 12) & 0x1) << 4 | ((hw >> 13) & 0x7) << 5 | xcc << 8
print("distinct CUs used", len(np.unique(cu)), "items per CU: min %d max %d" % (np.bincount(np.unique(cu, return_inverse=True)[1]).min(),
                                                                                np.bincount(np.unique(cu, return_inverse=True)[1]).max()))
busy = np.zeros(int(e.max()) + 2)
for s, t in zip(b, e):
    busy[int(s):int(t) + 1] += 1
print("workgroups resident per 10 us:", " ".join("%d" % busy[i:i + 10].mean() for i in range(0, len(busy), 10)))
per_cu_busy = {}
for c, s, t in zip(cu, b, e):
    per_cu_busy[c] = per_cu_busy.get(c, 0.0) + (t - s)
v = np.array(list(per_cu_busy.values()))
print("per-CU busy us: mean %.1f min %.1f max %.1f" % (v.mean(), v.min(), v.max()))
print("per XCD: items, busy CU-us, first idle CU at, last end")
for k in sorted(np.unique(xcc)):
    mk = xcc == k
    ends = {}
    for c, t_ in zip(cu[mk], e[mk]):
        ends[c] = max(ends.get(c, 0.0), t_)
    print("  xcd %d: %3d items (blockIdx %% 8: %s)  busy %7.0f  CUs %d  earliest-idle %.0f  last %.0f" % (
        k, mk.sum(), ",".join(str(v) for v in sorted(set((np.nonzero(mk)[0] % 8).tolist()))), (e[mk] - b[mk]).sum(), len(ends),
        min(ends.values()), max(ends.values())))
order = np.argsort(e)[-12:]
print("last finishers: " + " ".join("L%d/%dr/%.0f-%.0f" % (level[i], nrec[i], b[i], e[i]) for i in order))
# corners per routing record on the multi-slice levels (how many of a sample's 8 corners share a slice)
idx = ops.hashgrid_indices(x, meta).long()
print("level  records/sample  share of records with 1..8 corners in their slice")
for l in range(L):
    size = meta.offsets[l + 1] - meta.offsets[l]
    if size <= 10240:
        continue
    sl = (idx[:, l, :] >> 13)
    same = (sl[:, :, None] == sl[:, None, :]).sum(2)                # corners in the slice of corner c
    first = torch.ones_like(sl, dtype=torch.bool)
    for c in range(1, 8):
        first[:, c] = (sl[:, :c] != sl[:, c:c + 1]).all(1)
    sizes = same[first]
    h = torch.bincount(sizes, minlength=9)[1:].float()
    print("%3d %10.2f      %s" % (l, first.sum().item() / M, " ".join("%.3f" % v for v in (h / h.sum()).tolist())))
