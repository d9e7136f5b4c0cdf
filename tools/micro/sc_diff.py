"""Scatter A/B by RESULT: the recorded hashgrid_bwd call of one headline step run by the in-tree library and by experiment
builds into cleared gradient tables; per level: entries that differ, entries that are zero in one and not in the other.
    python tools/micro/sc_diff.py scil"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
import replay  # noqa: E402
from mipsfusion_amd import _lib, synth  # noqa: E402
from mipsfusion_amd.graph import work_stream  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = work_stream(dev)
cfg = synth.config_headline()
model, frames, poses = bench.build_submap(cfg, dev, seed=0)
table, db, R = bench.build_ray_table(cfg, frames, dev)
idx_rows, idx_owner = bench.draw_index_sets(cfg, frames, db, R, 64)
loop = bench.MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev)
for _ in range(55):
    loop.step()
torch.cuda.synchronize()
base = _lib.lib()
rec = replay.Recorder(base)
_lib._lib = rec
loop.i = 4
loop.step()
torch.cuda.synchronize()
_lib._lib = base
call = [(n, a) for n, a in rec.calls if n == "mipsf_hashgrid_bwd"][0]
args = call[1][0]
blk = args._obj if hasattr(args, "_obj") else args.contents if hasattr(args, "contents") else args
n_params = model.embed_fn.params.numel()
grad = model.embed_fn.params.grad
assert grad.data_ptr() == blk.dparams, (grad.data_ptr(), blk.dparams)
meta = _lib.make_grid_meta(16, 2, cfg["grid"]["hash_size"], 16, float(2.0 ** (4 / 15)))
offs = [int(o) * 2 for o in meta.offsets[:17]]
libs = [("base", base)] + [(n, replay.open_lib(os.path.join(ROOT, "tools", "micro", f"libv_{n}.so"))) for n in sys.argv[1:]]
res = {}
for name, h in libs:
    outs = []
    for rep in range(3):
        grad.zero_()
        torch.cuda.synchronize()
        rc = getattr(h, call[0])(*call[1])
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(grad.detach().clone())
    res[name] = outs
    print(f"{name}: run-to-run identical: {torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])}; nonzero entries {int((outs[0] != 0).sum())}")
ref = res["base"][0]
if os.environ.get("SC_DIFF_DUMP"):
    os.makedirs(os.path.dirname(os.environ["SC_DIFF_DUMP"]), exist_ok=True)
    torch.save({"g": ref[offs[9]:offs[11]].cpu(), "x": torch.as_tensor(0)}, os.environ["SC_DIFF_DUMP"])
for name, outs in res.items():
    for rep, o in enumerate(outs):
        if name == "base" and rep == 0:
            continue
        d = o != ref
        zz = ((o == 0) != (ref == 0))
        print(f"{name}[{rep}] vs base[0]: {int(d.sum())} entries differ (max rel {float(((o - ref).abs() / ref.abs().max()).max()):.2e}), "
              f"{int(zz.sum())} zero in one only")
        per = [int(d[offs[l]:offs[l + 1]].sum()) for l in range(16)]
        print(f"     entries that differ, by level: {per}")
        if int(zz.sum()):
            ii = torch.nonzero(zz).flatten()[:10].tolist()
            for i in ii:
                lvl = max(l for l in range(16) if offs[l] <= i)
                print(f"     entry {i} (level {lvl}, entry {(i - offs[lvl]) // 2}, feature {i & 1}): base {float(ref[i]):.3e} {name} {float(o[i]):.3e}")
