"""Does the RandomOptimizer need six products?  

A sub-map is trained on the first frames of bench.py's sequence (GraphedSequence, reference cadence); then, for every one of
the following frames, ONE RandomOptimizer frame (iter_RO rounds, the reference's particle template) is run from the same
constant-velocity start pose with each decoder arithmetic -- f32 (the fp32-MFMA kernels: the reference's operand width),
bf16x6 (default), f16x3, f16 -- and the tracked poses are compared with the f32 kernels': bit equality, translation and
rotation difference, and the difference's size against the frame's own tracking error.  Prints a table and one JSON line.

usage: python tools/ro_arithmetic.py [--train-frames 16] [--frames 40]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (first: host CPU confinement and OMP_NUM_THREADS before torch is imported)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipsfusion_amd import sequence, synth  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from run_sequence import trajectory  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--train-frames", type=int, default=16)
ap.add_argument("--frames", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
torch.manual_seed(0), np.random.seed(0)
cfg = synth.config_reference_defaults()
n_all = args.train_frames + args.frames
gt = trajectory(cfg, n_all)
frames = [synth.make_frame(cfg, gt[k], seed=k, frame_id=k) for k in range(n_all)]
t0 = time.perf_counter()
with torch.cuda.stream(stream):
    seq = sequence.GraphedSequence(cfg, dev, frames[:args.train_frames], kf_every=15, sampler="reference", stream=stream)
    res = seq.run(gt[:args.train_frames])
torch.cuda.synchronize()
print(f"sub-map trained on {args.train_frames} frames in {time.perf_counter() - t0:.1f}s", file=sys.stderr)
model, ro = seq.model, seq.ro
model.eval()
n_ro = cfg["tracking"]["iter_RO"]
KINDS = ("f32", "bf16x6", "f16x3", "f16")


def angle_deg(Ra, Rb):
    # small-angle form: |Ra - Rb|_F = 2 sqrt(2) sin(angle / 2); exactly 0 for equal matrices (acos of the trace is not)
    return float(torch.rad2deg(2.0 * torch.asin(((Ra.double() - Rb.double()).norm() / (2.0 * 2.0 ** 0.5)).clamp(max=1.0))))


rows = {k: {"equal": 0, "dt_mm": [], "dr_deg": [], "err_mm": []} for k in KINDS}
moved = 0          # frames in which the f32 rounds moved the pose at all (a frame whose start pose wins is equal trivially)
with torch.cuda.stream(stream):
    for k in range(args.train_frames, n_all):
        # constant-velocity prediction from the ground truth of the two frames before (mipsfusion.py:470-478)
        prev, prev2 = gt[k - 1].float(), gt[k - 2].float()
        init = (prev @ torch.linalg.inv(prev2) @ prev).float()
        poses = {}
        for kind in KINDS:
            ro.decoder_precision = kind
            poses[kind] = ro.optimize(model, frames[k]["depth"], init, None, n_iter=n_ro).cpu()
        moved += int(not torch.equal(poses["f32"], init))
        for kind in KINDS:
            p, q = poses[kind], poses["f32"]
            rows[kind]["equal"] += int(torch.equal(p, q))
            rows[kind]["dt_mm"].append(float((p[:3, 3] - q[:3, 3]).norm()) * 1e3)
            rows[kind]["dr_deg"].append(angle_deg(p[:3, :3], q[:3, :3]))
            rows[kind]["err_mm"].append(float((p[:3, 3] - gt[k][:3, 3].float()).norm()) * 1e3)
out = {}
print(f"{'arithmetic':10s} {'= f32 pose':>11s} {'|dt| mean':>10s} {'max [mm]':>9s} {'rot mean':>9s} {'max [deg]':>9s} {'error to truth mean [mm]':>25s}")
for kind in KINDS:
    r = rows[kind]
    out[kind] = {"frames": args.frames, "pose_bit_equal_to_f32": r["equal"], "dt_mm_mean": round(float(np.mean(r["dt_mm"])), 5),
                 "dt_mm_max": round(float(np.max(r["dt_mm"])), 5), "drot_deg_mean": round(float(np.mean(r["dr_deg"])), 6),
                 "drot_deg_max": round(float(np.max(r["dr_deg"])), 6), "err_to_truth_mm_mean": round(float(np.mean(r["err_mm"])), 3)}
    o = out[kind]
    print(f"{kind:10s} {o['pose_bit_equal_to_f32']:5d}/{args.frames:<5d} {o['dt_mm_mean']:10.5f} {o['dt_mm_max']:9.5f} {o['drot_deg_mean']:9.6f} "
          f"{o['drot_deg_max']:9.6f} {o['err_to_truth_mm_mean']:25.3f}")
print(f"frames in which the f32 rounds moved the start pose: {moved} of {args.frames}")
print(json.dumps({"ro_arithmetic_vs_f32": out, "rounds_per_frame": n_ro, "frames_moved_by_f32": moved}))
sys.stdout.flush()
os._exit(0)
