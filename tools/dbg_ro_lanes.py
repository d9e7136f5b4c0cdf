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
other = None
if "--beside" in sys.argv:        # a second process that runs local-BA mapping steps on the same GPU (persistent kernels that
    import subprocess             # hold a CU's whole LDS: the two processes are time-sliced)
    other = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "ba_load.py"), "--seconds", "200"],
                             stdout=subprocess.PIPE, text=True, env={k: v for k, v in os.environ.items() if k != "MIPSF_LIB"})
    assert other.stdout.readline().strip() == "READY"
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
cnt = {"xn": 0, "pst7": 0, "feat": 0, "sdf": 0, "fit": 0}
detail = []
faulty = []
for k in range(N):
    xn, pst7 = ops.ro_particles(ro.pre_sampled_particle, state.clone(), ro._dirs[0], td, rc, point_major=_POINT_MAJOR)
    cur = {"xn": xn, "pst7": pst7}
    if load:
        feat = ops.hashgrid_fwd(xn, model.embed_fn.params.detach(), model.embed_fn.meta, FEAT_LEVEL_MAJOR)
        sdf = ops.decoder_fwd_sdf(None, feat, FEAT_LEVEL_MAJOR, xn, None, xn.shape[0], precision="f16x3", packed16=pk)
        fit = ops.ro_fitness(sdf.view(ro.particle_size, -1, 1), td, ro.trunc_value, point_major=_POINT_MAJOR)
        cur.update(feat=feat, sdf=sdf, fit=fit)
    if first is None:
        first = {a: b.clone() for a, b in cur.items()}
        continue
    for a, b in cur.items():
        if not torch.equal(b, first[a]):
            cnt[a] += 1
            if len(detail) < 8:
                d = (b.float() - first[a].float()).abs().reshape(-1)
                nz = (d > 0).nonzero().reshape(-1)
                detail.append(f"launch {k}: {a}: {int(nz.numel())} elements differ (max {float(d.max()):.3e}), flat indices {nz[:6].tolist()} .. {int(nz[-1])}")
            if a == "xn" and len(faulty) < 64:
                faulty.append((k, b.clone()))
torch.cuda.synchronize()
print(f"pid {os.getpid()}{' beside ba_load' if other else ''}: of {N} launches, stage outputs differing from the first launch's: {cnt}")
for ln in detail:
    print("   ", ln)
if other is not None:
    other.terminate()
if faulty:
    # what the wrong lanes computed: undo the normalisation (world = xn * norm_factor * div + sub) and fit the difference from the
    # right world coordinate as  da . c + dt  over the faulty lanes (c = the lattice point in camera coordinates): a wave-uniform
    # wrong operand (a rotation row or the translation) fits exactly, a per-lane arithmetic error does not
    P, n = ro.pre_sampled_particle.shape[0], td.shape[0]
    sub = np.array([-rc.half_len[d] for d in range(3)]) if not rc.use_bound else np.array([rc.bound_min[d] for d in range(3)])
    div = np.array([2 * rc.half_len[d] for d in range(3)]) if not rc.use_bound else np.array([rc.bound_max[d] - rc.bound_min[d] for d in range(3)])
    nf = rc.norm_factor
    cam = (ro._dirs[0].double() * td.double()[:, None]).cpu().numpy()             # [n,3]
    good = first["xn"].double().cpu().numpy().reshape(n, P, 3) if _POINT_MAJOR else first["xn"].double().cpu().numpy().reshape(P, n, 3).transpose(1, 0, 2)
    world_good = good * nf * div + sub                                            # [n,P,3]
    print(f"  P {P} particles, n {n} lattice points, norm sub {sub.tolist()} div {div.tolist()} factor {nf}")
    for k, b in faulty[:24]:
        bad = b.double().cpu().numpy().reshape(n, P, 3) if _POINT_MAJOR else b.double().cpu().numpy().reshape(P, n, 3).transpose(1, 0, 2)
        ii, pp, cc = np.nonzero(bad != good)
        world_bad = bad * nf * div + sub
        for p_ in sorted(set(pp.tolist())):
            for c_ in sorted(set(cc[pp == p_].tolist())):
                pts = np.sort(ii[(pp == p_) & (cc == c_)])
                dw = world_bad[pts, p_, c_] - world_good[pts, p_, c_]
                A = np.concatenate([cam[pts], np.ones((len(pts), 1))], axis=1)
                sol, res, *_ = np.linalg.lstsq(A, dw, rcond=None)
                fit_err = np.abs(A @ sol - dw).max()
                # does another particle's pose give these values?
                other_match = [int(q) for q in range(P) if np.abs(world_good[pts, q, c_] - world_bad[pts, p_, c_]).max() < 1e-6]
                other_comp = [c2 for c2 in range(3) if np.abs(world_good[pts, p_, c2] - world_bad[pts, p_, c_]).max() < 1e-6]
                Aall = np.concatenate([cam, np.ones((n, 1))], axis=1)
                rows = [np.linalg.lstsq(Aall, world_good[:, p_, c2], rcond=None)[0] for c2 in range(3)]   # the particle's aR rows | t
                print(f"      the particle's pose: row0 {np.round(rows[0], 5).tolist()} row1 {np.round(rows[1], 5).tolist()} row2 {np.round(rows[2], 5).tolist()}")
                print(f"    launch {k}: particle {p_} (wave {p_ % 4} of block {p_ // 4}), component {c_}, points {pts[0]}..{pts[-1]} ({len(pts)}): "
                      f"world diff {dw.min():+.4f}..{dw.max():+.4f}; fit da {np.round(sol[:3], 5).tolist()} dt {sol[3]:+.5f} max error {fit_err:.2e}; "
                      f"equals particle(s) {other_match} same component; equals own component(s) {other_comp}")
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
