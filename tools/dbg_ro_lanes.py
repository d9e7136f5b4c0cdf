"""Item 'lanes 48..63': run the RandomOptimizer's particle kernel of a -DMIPSF_RO_LANE_CHECK build many times (two instances of
this script at once share the GPU) and print, for every wavefront whose lanes left the pose section with different values,
which intermediate differs first.  Build: tools/micro/variant.sh rolanes ro -DMIPSF_RO_LANE_CHECK;
run: MIPSF_LIB=$PWD/tools/micro/libv_rolanes.so python tools/dbg_ro_lanes.py [rounds] [--load]   (--load: the full round --
grid lookups + decoder -- between the particle kernels, as in a frame)"""
import ctypes as C
import os
import sys
import types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from mipsfusion_amd import _lib, ops, synth
from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR
from mipsfusion_amd.RandomOptimizer import RandomOptimizer, _POINT_MAJOR

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3000
load = "--load" in sys.argv
dev = torch.device("cuda:0")
cfg = synth.config_headline()
cfg["tracking"]["RO"].setdefault("initial_scaling_factor", 0.02)
cfg["tracking"]["RO"].setdefault("rescaling_factor", 0.5)
cfg["tracking"].setdefault("ignore_edge_W", 20)
cfg["tracking"].setdefault("ignore_edge_H", 20)
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
model.eval()
H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
f = synth.make_frame(cfg, seed=1)
ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=f["direction"])
np.random.seed(0)
ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
init = f["c2w"].clone()
init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
td = f["depth"][ro.row_indices, ro.col_indices].to(dev, torch.float32).contiguous()
state = torch.zeros(ops.RO_STATE_FLOATS, device=dev)
state[0:9], state[9:12], state[12:18] = init[:3, :3].reshape(9).to(dev), init[:3, 3].to(dev), 0.02
rc = model._rc(1, 0)
ws = model.decoder.ordered_parameters()
pk = ops.decoder_pack16(ws, precision="f16x3")
first = None
bad = 0
for k in range(N):
    xn, pst7 = ops.ro_particles(ro.pre_sampled_particle, state.clone(), ro._dirs[0], td, rc, point_major=_POINT_MAJOR)
    if load:
        feat = ops.hashgrid_fwd(xn, model.embed_fn.params.detach(), model.embed_fn.meta, FEAT_LEVEL_MAJOR)
        ops.decoder_fwd_sdf(None, feat, FEAT_LEVEL_MAJOR, xn, None, xn.shape[0], precision="f16x3", packed16=pk)
    if first is None:
        first = xn.clone()
    elif not torch.equal(xn, first):
        bad += 1
torch.cuda.synchronize()
print(f"pid {os.getpid()}: {bad} of {N} launches produced points that differ from the first launch's")
lib = _lib.lib()
if hasattr(lib, "mipsf_ro_chk_read"):
    cnt = C.c_uint(0)
    dump = np.zeros((32, 64, 24), dtype=np.float32)
    lib.mipsf_ro_chk_read.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.mipsf_ro_chk_read(C.byref(cnt), dump.ctypes.data) == 0
    print(f"  in-kernel check: {cnt.value} wavefronts left the pose section with lanes that differ from lane 0")
    names = ["r0", "r1", "r2", "r3", "r4", "r5", "s", "qw"] + [f"aR{k}" for k in range(9)] + ["t0", "t1", "t2", "pst0", "search0", "dR0", "p"]
    for slot in range(min(cnt.value, 6)):
        d = dump[slot]
        bits = d.view(np.uint32)
        diff_lanes = [l for l in range(64) if (bits[l, :20] != bits[0, :20]).any()]
        first_cols = sorted({int(np.nonzero(bits[l, :24] != bits[0, :24])[0][0]) for l in diff_lanes})
        print(f"  fault {slot}: particle {int(d[0, 23])}, lanes {diff_lanes[0]}..{diff_lanes[-1]} ({len(diff_lanes)}); first differing "
              f"column(s): {[names[c] for c in first_cols]}")
        l = diff_lanes[0]
        for c in range(23):
            if bits[l, c] != bits[0, c]:
                print(f"      {names[c]:8s} lane 0 {d[0, c]: .9e} (0x{bits[0, c]:08x})   lane {l} {d[l, c]: .9e} (0x{bits[l, c]:08x})")
else:
    print("  (library without -DMIPSF_RO_LANE_CHECK)")
