"""Kernel A/B on a FIXED workload: records every C-ABI call of one headline mapping step (BASELINE config 2, after the
usual set-up iterations) and replays the recorded sequence -- same pointers, same contents -- against the in-tree library
and any number of experiment builds (tools/micro/variant.sh), timing each call with an event pair on the launch stream.

    python tools/replay.py [name ...]         names of tools/micro/libv_<name>.so; "base" is always measured first
    REPLAY_ONLY=regex                         replay (and time) only the calls whose entry point matches
    REPLAY_REPS=20

The optimiser calls are left out of the replay, so the parameters -- and with them every tensor a kernel reads -- stay what
they were when the step was recorded: a diagnosis build that produces garbage (loads only, compute only) cannot change the
work of the kernels that run after it.  (bench.py's own per-kernel table feeds back: with a loads-only weight-gradient
kernel the model diverges and the live-tile share of the next steps changes.)
"""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "16")
import torch  # noqa: E402

import bench  # noqa: E402
from mipsfusion_amd import _lib, ops, synth  # noqa: E402
from mipsfusion_amd.graph import work_stream  # noqa: E402


class Recorder:
    def __init__(self, handle):
        self.handle, self.calls = handle, []

    def __getattr__(self, name):
        fn = getattr(self.handle, name)
        if not name.startswith("mipsf_") or fn.restype is not C.c_int:
            return fn

        def wrapped(*args):
            self.calls.append((name, args))
            return fn(*args)
        return wrapped


def open_lib(path):
    h = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    return h


def report_wgrad_trace(lname, h):
    """-DW16_TRACE builds: cycles per tile between the marks of the exchange roles of wgrad16.hip, per role."""
    try:
        fn = h.mipsf_w16_trace_read
    except AttributeError:
        return
    import numpy as np
    buf = np.zeros(2048 * 16, dtype=np.uint64)
    fn(buf.ctypes.data_as(C.c_void_p), 1)
    t = buf.reshape(2048, 16).astype(np.float64)
    names = {"A": ["H1[w] -> XA", "e(next)[w] -> XE", "X = dH2[w]^T", "barrier", "4 x mac", "small rows + next loads", "second barrier"],
             "B": ["column tile -> XB", "X3 = dG3^T (or: the next records' loads)", "X1 = dG1^T", "barrier", "5 x mac", "next tile's X3, X1 (early form)",
                   "second barrier"]}
    for role, sel in (("A", lambda w: w % 8 < 4), ("B", lambda w: w % 8 >= 4)):
        rows = np.array([w for w in range(2048) if sel(w) and t[w, 15] > 0])
        if rows.size == 0:
            continue
        per = t[rows, :7].sum(0) / t[rows, 15].sum()
        print(f"  [{lname}] role {role}: {per.sum():.0f} cycles per tile: " + ", ".join(f"{n} {c:.0f}" for n, c in zip(names[role], per)))
        for k in range(4):
            r2 = rows[rows % 4 == k]
            p2 = t[r2, :7].sum(0) / t[r2, 15].sum()
            print(f"      wave {k + (0 if role == 'A' else 4)}: " + " ".join(f"{c:6.0f}" for c in p2))


def report_chain_trace(lname, h, calls):
    """-DD16_TRACE builds: cycles per LIVE tile between the marks of the backward chain's tile (one more pass over the calls,
    the trace cleared in front of the chain call and read behind it)."""
    try:
        fn = h.mipsf_d16_trace_read
    except AttributeError:
        return
    import numpy as np
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    for n, a in calls:
        if "bwd_chain" in n:
            torch.cuda.synchronize()
            fn(buf.ctypes.data_as(C.c_void_p), 1)
        getattr(h, n)(*a)
        if "bwd_chain" in n:
            torch.cuda.synchronize()
            fn(buf.ctypes.data_as(C.c_void_p), 1)
            t = buf.reshape(4096, 16).astype(np.float64)
            used = t[:, 15] > 0
            names = ["dout loads + liveness", "out / masks / x loads", "softmax backward + scales", "S2T + mask", "B3 (+ dG3 stores)",
                     "unscale + dfeat stores", "RGBT + B2 (+ dH2 stores)", "mask + B1 (+ dG1 stores)", "de -> dx"]
            per = t[used, :9].sum(0) / t[used, 15].sum()
            ns = t[used, 12].sum() / t[used, 15].sum() * 10.0
            print(f"  [{lname}] chain: {int(t[used, 15].sum())} live tiles, {per.sum():.0f} cycles per tile in {ns:.0f} ns ({per.sum() / ns:.2f} GHz), "
                  f"tiles per wave {t[used, 15].mean():.2f} (max {t[used, 15].max():.0f})")
            for nm, c in zip(names, per):
                print(f"      {nm:34s} {c:7.0f} {100 * c / per.sum():5.1f} %")
    torch.cuda.synchronize()


def main():
    names = [a for a in sys.argv[1:]]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = work_stream(dev)
    cfg = synth.config_headline()
    model, frames, poses = bench.build_submap(cfg, dev, seed=0)
    table, db, R = bench.build_ray_table(cfg, frames, dev)
    idx_rows, idx_owner = bench.draw_index_sets(cfg, frames, db, R, 64)
    loop = bench.MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev)
    for _ in range(int(os.environ.get("REPLAY_SETUP", "55"))):
        loop.step()
    torch.cuda.synchronize()
    base = _lib.lib()
    rec = Recorder(base)
    _lib._lib = rec
    loop.i = 4                      # a step that also runs the pose optimiser (pose_accum_step = 5)
    loss = loop.step()
    torch.cuda.synchronize()
    _lib._lib = base
    live = ops.last_live_tile_share()
    calls = [(n, a) for n, a in rec.calls if "adam" not in n]
    only = os.environ.get("REPLAY_ONLY")
    if only:
        calls = [(n, a) for n, a in calls if re.search(only, n)]
    print(f"recorded {len(rec.calls)} calls, replaying {len(calls)}; loss {float(loss):.5f}; live tile share {live}")
    reps = int(os.environ.get("REPLAY_REPS", "20"))
    # "base" (the in-tree library) is always measured first; naming it again ("base2") measures it once more in that position --
    # the drift of the box over the run
    libs = [("base", base)] + [(n, base if n.startswith("base") else open_lib(os.path.join(ROOT, "tools", "micro", f"libv_{n}.so")))
                               for n in names if n != "base"]
    table_out = {}
    for lname, h in libs:
        evs = [[] for _ in calls]
        for r in range(reps + 3):
            for k, (n, a) in enumerate(calls):
                fn = getattr(h, n)
                if r >= 3:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                rc = fn(*a)
                if r >= 3:
                    e1.record()
                    evs[k].append((e0, e1))
                if rc != 0:
                    raise RuntimeError(f"{n}: rc {rc}: {h.mipsf_last_error().decode()}")
        torch.cuda.synchronize()
        for k, (n, _a) in enumerate(calls):
            us = sum(a.elapsed_time(b) for a, b in evs[k]) / len(evs[k]) * 1e3
            table_out.setdefault((k, n), {})[lname] = us
        report_wgrad_trace(lname, h)
        report_chain_trace(lname, h, calls)
    if os.environ.get("REPLAY_CLOCK"):
        # the board's state under each replayed call alone: the call in a loop for REPLAY_CLOCK seconds (in-tree library), shader
        # clock and package power sampled beside it (bench.BoardSampler: sysfs pp_dpm_sclk / hwmon) -- is the kernel at the
        # board's power cap?  (tools/clock_probe.py does this for the decoder forward's forms.)
        secs = float(os.environ["REPLAY_CLOCK"])
        import time
        print(f"board state under each call alone ({secs:.1f} s loops):")
        with bench.BoardSampler(0, period=0.02) as b0:
            time.sleep(1.0)
        print(f"  {'idle':34s} {b0.summary()}")
        for k, (n, a) in enumerate(calls):
            fn = getattr(base, n)
            # (the chain finds its live-tile counters cleared by the forward of the same record: alone in a loop it would
            # process nothing after the first call -- loop the pair and say so)
            pre = [(getattr(base, m), b) for m, b in calls[:k] if "fwd16" in m][-1:] if "bwd_chain" in n else []
            label = n[6:] + (" (+ fwd16 in front)" if pre else "")
            torch.cuda.synchronize()
            t0, reps = time.perf_counter(), 0
            with bench.BoardSampler(0, period=0.02) as bs:
                while time.perf_counter() - t0 < secs:
                    for _ in range(50):
                        for pf, pa in pre:
                            pf(*pa)
                        fn(*a)
                    reps += 50
                    torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / reps * 1e6
            sm = bs.summary()
            print(f"  {k:2d} {label:38s} {us:8.1f} us/call  sclk median {sm['sclk_mhz_median']} min {sm['sclk_mhz_min']} MHz, "
                  f"power median {sm['power_w_median']} W ({sm['samples']} samples)")
    w = max(len(n) for _, n in table_out) + 4
    print(" " * w + "".join(f"{l:>12s}" for l, _ in libs))
    tot = {l: 0.0 for l, _ in libs}
    for (k, n), row in sorted(table_out.items()):
        print(f"{k:2d} {n[6:]:{w - 3}s}" + "".join(f"{row[l]:12.1f}" for l, _ in libs))
        for l, _ in libs:
            tot[l] += row[l]
    print(f"{'sum':{w}s}" + "".join(f"{tot[l]:12.1f}" for l, _ in libs))


if __name__ == "__main__":
    main()
