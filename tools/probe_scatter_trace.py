"""Timeline of the routed scatter's work items on the headline step's own (x, dL/dy): needs a library built with
-DMIPSF_SC_TRACE (tools/micro/variant.sh trace hashgrid -DMIPSF_SC_TRACE; MIPSF_LIB=tools/micro/libv_trace.so).
Prints, per level: items, records, time per item; per workgroup: busy time and items; the kernel's span."""
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
lib.mipsf_hashgrid_trace_words.restype = C.c_uint64
lib.mipsf_hashgrid_trace_words.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
n_rows = C.c_uint32()
bin0 = (C.c_uint32 * 33)()
off = lib.mipsf_hashgrid_trace_words(C.byref(meta), M, C.byref(n_rows), bin0)
n = _lib.buffer_size(_lib.SIZE_HASHGRID_BWD_SCRATCH, M, 0, 0, meta)
scratch = torch.zeros(n, dtype=torch.float32, device=dev)
dparams = torch.zeros_like(params)
ARGS = _lib.HashgridBwdArgs.new(M=M, x=ops.dptr(x), params=ops.dptr(params), dout=ops.dptr(dout), dparams=ops.dptr(dparams),
                                scratch=ops.dptr(scratch), meta=C.pointer(meta), feat_layout=layout)
L = meta.n_levels
for rep in range(3):
    scratch[off - 1:off + 8 * n_rows.value].zero_()
    ops.check(lib.mipsf_hashgrid_bwd(C.byref(ARGS), ops.stream_ptr()), "bwd")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for rep in range(20):
    lib.mipsf_hashgrid_bwd(C.byref(ARGS), ops.stream_ptr())
e1.record()
torch.cuda.synchronize()
print(f"whole scatter (clear + route + accumulate + fold, with trace stores): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
total = int(scratch[off - 1:off].view(torch.int32).item())
tr = scratch[off:off + 8 * total].view(torch.int32).cpu().numpy().view(np.uint32).reshape(-1, 8)
t0 = tr[:, 3].astype(np.int64)
t1 = tr[:, 4].astype(np.int64)
dur = ((t1 - t0) % (1 << 32)) * 0.01            # us (100 MHz)
start = ((t0 - t0.min()) % (1 << 32)) * 0.01
end = start + dur
bins = tr[:, 1] & 0xffff
level = np.searchsorted(np.array(bin0[:L + 1]), bins, side="right") - 1
print(f"{total} items; span {end.max():.1f} us; sum of item times {dur.sum():.0f} us = {dur.sum() / 256:.1f} us per CU")
PART = int(os.environ.get("PART", "4096"))
parts_of = tr[:, 7].astype(np.int64)
recs_item = np.minimum(tr[:, 2].astype(np.int64) - (tr[:, 1] >> 16).astype(np.int64) * PART, PART)
print(f"records {recs_item.sum()}; single-part items {(parts_of == 1).sum()}, items of split bins {(parts_of > 1).sum()}")
for l in range(L):
    m = level == l
    if m.any():
        print(f"  level {l:2d}: {m.sum():4d} items, records {recs_item[m].sum():9d}, "
              f"per item: records {recs_item[m].mean():7.0f} time mean {dur[m].mean():6.1f} max {dur[m].max():6.1f} us (sum {dur[m].sum():7.0f}), "
              f"started {start[m].min():6.1f}..{start[m].max():6.1f}")
wg = tr[:, 6]
W = int(wg.max()) + 1
busy = np.bincount(wg, weights=dur, minlength=W)
nitem = np.bincount(wg, minlength=W)
last = np.array([end[wg == w].max() if (wg == w).any() else 0 for w in range(W)])
first = np.array([start[wg == w].min() if (wg == w).any() else 0 for w in range(W)])
print(f"per workgroup: items {nitem.min()}..{nitem.max()}, busy {busy.min():.1f}..{busy.max():.1f} (mean {busy.mean():.1f}) us, "
      f"first item starts {first.min():.1f}..{first.max():.1f}, last item ends {last.min():.1f}..{last.max():.1f}")
gaps = (last - first - busy)
print(f"time between a workgroup's items (dequeue + lookup + barriers): mean {gaps.mean():.1f} us per workgroup, {gaps.sum() / max(1, (nitem - 1).clip(0).sum()):.2f} us per hand-over")
xcc = tr[:, 5] & 0xf
for k in range(8):
    m = xcc == k
    print(f"  XCC {k}: {m.sum():3d} items, {dur[m].sum():7.0f} us")
