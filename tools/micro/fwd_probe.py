"""Forward decoder kernel alone (MIPSF_PROBE_PREC: bf16x6 | f16x3 | f16) at the headline batch: the evaluation form and the training form (lean record).
MIPSF_LIB selects an experiment library (tools/micro/variant.sh <name> decoder16 -DD16_ABL=<bits>)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from mipsfusion_amd import ops, synth
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
M = int(os.environ.get("MIPSF_PROBE_M", "262144"))
PREC = os.environ.get("MIPSF_PROBE_PREC", "bf16x6")
packed16 = ops.decoder_pack16(model.decoder.ordered_parameters(), precision=PREC)
x = torch.rand(M, 3, device=dev)
feat = torch.randn(16, M, 2, device=dev) * 1e-2
import ctypes as C
import numpy as np
from mipsfusion_amd import _lib
PHASES = ["e (sin)", "layer 1", "relu 1 + masks", "layer 2 (+H1 stores)", "unscale 2 + grid loads", "rgb head", "rgb_emb stores",
          "layer 3 (+H2 stores)", "relu 3 + H3 stores + masks", "sdf head", "softmax + out"]
tracer = getattr(C.CDLL(_lib.LIB_PATH), "mipsf_d16_trace_read", None) if os.environ.get("MIPSF_LIB") else None
for save in (False, "lean", "masks"):
    fn = lambda: ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save=save, precision=PREC, packed16=packed16)   # noqa: E731
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{os.path.basename(os.environ.get('MIPSF_LIB', 'base')):28s} save={str(save):5s} {a.elapsed_time(b) / 20 * 1e3:7.1f} us")
    if tracer is not None:
        buf = np.zeros(4096 * 16, dtype=np.uint64)
        tracer(buf.ctypes.data_as(C.c_void_p), 1)                   # everything so far (warm-up + timed): cleared
        fn()
        torch.cuda.synchronize()
        tracer(buf.ctypes.data_as(C.c_void_p), 1)
        t = buf.reshape(4096, 16)
        used = t[:, 15] > 0
        per_tile = t[used, :11].sum(0) / t[used, 15].sum()
        ns = t[used, 12].sum() / t[used, 15].sum() * 10.0
        print(f"    cycles per tile-wave (s_memtime = shader cycles, mean over {int(t[used, 15].sum())} tiles): total {per_tile.sum():.0f} "
              f"in {ns:.0f} ns => {per_tile.sum() / ns:.2f} GHz")
        for name, c in zip(PHASES, per_tile):
            print(f"      {name:32s} {c:8.0f}  {100 * c / per_tile.sum():5.1f} %")
        k0, k1, kend = t[used, 13].astype(np.int64), t[used, 14].astype(np.int64), t[used, 11].astype(np.int64)
        z = k0.min()
        print(f"    wall clock (us, from the first wave's entry): entry {((k0 - z) * 0.01).min():.2f}..{((k0 - z) * 0.01).max():.2f}, "
              f"images in LDS at {((k1 - z) * 0.01).min():.2f}..{((k1 - z) * 0.01).max():.2f} (mean {((k1 - z) * 0.01).mean():.2f}), "
              f"last tile done at {((kend - z) * 0.01).min():.2f}..{((kend - z) * 0.01).max():.2f} (mean {((kend - z) * 0.01).mean():.2f})")
        wid = np.nonzero(used)[0]
        xcd = (wid // 8) % 8
        tile_ns = t[used, 12] / t[used, 15] * 10.0
        cyc = t[used, :11].sum(1) / t[used, 15]
        print("    per XCD: " + "  ".join(f"{k}: {tile_ns[xcd == k].mean() / 1e3:.2f} us/tile {cyc[xcd == k].mean() / tile_ns[xcd == k].mean():.2f} GHz" for k in range(8)))
        cu_ns = np.array([tile_ns[(wid // 8) == c].mean() for c in np.unique(wid // 8)])
        print(f"    per workgroup (CU) mean us/tile: min {cu_ns.min() / 1e3:.2f} p10 {np.percentile(cu_ns, 10) / 1e3:.2f} median {np.median(cu_ns) / 1e3:.2f} "
              f"p90 {np.percentile(cu_ns, 90) / 1e3:.2f} max {cu_ns.max() / 1e3:.2f}; within a workgroup (max - min over its 8 waves): "
              f"{np.mean([np.ptp(tile_ns[(wid // 8) == c]) for c in np.unique(wid // 8)]) / 1e3:.2f} us")
