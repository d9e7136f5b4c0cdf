"""Does the lanes-48..63 fault of the PACKED build of ro_particles_kernel need a second PROCESS, or only the decoder's
persistent kernels on the same CUs?  One process, two streams: the recorded decoder calls of a mapping step replayed on the
stream they were recorded on, the particle kernel launched beside them on another stream and compared launch by launch.
    tools/micro/variant.sh ropk1 ro -DMIPSF_RO_PACKED=1 -DMIPSF_KEEP_PACKED_FP32      (the kernel as rounds 3-4 shipped it)
    MIPSF_LIB=$PWD/tools/micro/libv_ropk1.so python tools/dbg_ro_inproc.py [seconds] [regex of the calls to run beside]
    (PREC=f16x3|f32: the decoder arithmetic of the neighbour; QUICK=1: one alone/beside pair)"""
import os
import re
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench
from mipsfusion_amd import _lib, ops, synth
from replay import Recorder

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
only = sys.argv[2] if len(sys.argv) > 2 else "decoder_chain|decoder_fwd|decoder_bwd"
dev = torch.device("cuda:0")
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=5)
model.decoder_precision = os.environ.get("PREC", model.decoder_precision)
table, db, R = bench.build_ray_table(cfg, frames, dev)
rows, owner = bench.draw_index_sets(cfg, frames, db, R, 32)
loop = bench.MappingLoop(cfg, model, poses, table, rows, owner, dev)
for _ in range(5):
    loop.step()
torch.cuda.synchronize()
base = _lib.lib()
rec = Recorder(base)
_lib._lib = rec
loop.i = 4
loop.step()
torch.cuda.synchronize()
_lib._lib = base
calls = [(n, a) for n, a in rec.calls if "adam" not in n and re.search(only, n)]
P, n = 2000, 384
g = torch.Generator().manual_seed(1)
pst = (torch.rand(P, 6, generator=g) * 2 - 1).to(dev)
pst[0] = 0
state = torch.zeros(ops.RO_STATE_FLOATS, device=dev)
state[:12] = torch.tensor([0.962, -0.059, 0.266, 0.011, 0.984, 0.178, -0.272, -0.169, 0.947, 1.168, 3.796, 0.946])
state[12:18] = 0.02
dirs = torch.stack([torch.rand(n, generator=g) - 0.5, 0.8 * (torch.rand(n, generator=g) - 0.5), torch.ones(n)], 1).contiguous().to(dev)
depth = (0.8 + 2.2 * torch.rand(n, generator=g)).to(dev)
rc = model._rc(1, 0)
side = torch.cuda.Stream(dev)
with torch.cuda.stream(side):
    while True:
        ref, _ = ops.ro_particles(pst, state, dirs, depth, rc, point_major=True)
        again, _ = ops.ro_particles(pst, state, dirs, depth, rc, point_major=True)
        if torch.equal(ref, again):
            break
for beside in ((False, True) if os.environ.get("QUICK") else (False, True, False, True)):
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    launches, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        if beside:
            for _ in range(4):
                for name, a in calls:
                    assert getattr(base, name)(*a) == 0, name
        with torch.cuda.stream(side):
            for _ in range(32):
                xn, _ = ops.ro_particles(pst, state, dirs, depth, rc, point_major=True)
                bad += (xn != ref).any()
                launches += 1
        torch.cuda.synchronize()
    print(f"{'beside ' + str(sorted({c[0][6:] for c in calls})) + ' on another stream of this process' if beside else 'alone'}: "
          f"{int(bad)} of {launches} launches differ", flush=True)
