"""BASELINE config 3 on its own: python tools/run_config3.py [frames] [sampler] -> one JSON line (see bench.measured_config3)."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (sets the OpenMP / NUMA environment before torch is imported)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipsfusion_amd import sequence, synth  # noqa: E402
from mipsfusion_amd.graph import work_stream  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sampler = sys.argv[2] if len(sys.argv) > 2 else "reference"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
stream = work_stream(dev)
random.seed(0), np.random.seed(0), torch.manual_seed(0)
cfg = synth.config_two_rooms()
t0 = time.perf_counter()
gt, frames, schedule = synth.two_room_sequence(cfg, n_frames)
print(f"rendered {n_frames} frames in {time.perf_counter() - t0:.1f}s; schedule {schedule}", file=sys.stderr)
seq = sequence.GraphedSequence(cfg, dev, frames, kf_every=15, sampler=sampler, stream=stream, schedule=schedule)
res = seq.run(gt)
out = sequence.summarise(res, gt, cfg, "graphs")
err = [float((res["est"][k][:3, 3].float() - gt[k][:3, 3].float()).norm()) for k in range(len(res["est"]))]
out["err_every_15"] = [round(e, 4) for e in err[::15]]
out.pop("frame_ms_all", None)
out["err_all_cm"] = [round(e * 100, 1) for e in err]
print(json.dumps(out))
