#!/usr/bin/env python3
"""Benchmark of the render-and-optimise hot path (BASELINE.json metric) on 1..N MI355X GPUs.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full mapping iteration of MIPSFusion.local_BA (mipsfusion.py:293-342) at BASELINE config 2:
4096 rays x 64 samples (43 uniform + 21 depth-guided), hash grid 2^19, apartment_2 bound, 620x460 synthetic RGB-D:
ray build from the keyframe pose Parameters -> sample placement -> hash grid -> decoder -> SDF compositing ->
4 losses -> backward (grid, decoder, pose gradients) -> dense map Adam (+ pose Adam every pose_accum_step).
Every step processes a FRESH ray batch: the keyframe ray database (4 keyframes x 30 000 rays) and the current frame
live in one HBM table, the index sets are drawn on the host BEFORE the timed region with the reference's own samplers
(python random.sample per keyframeSet.py:386-436, sample_pixels_mix for the current frame) and resident in HBM with
the jitter when timing starts; the rows are gathered inside the step.  N > 1: one process per GPU, each optimising its
OWN submap (weak scaling); the only exchange of the timed region is an all-gather of the optimised keyframe
poses once per BA round (mapping.iters steps); the other shardings are exercised after it (`multi_gpu`).

Besides the contract fields the line carries: `step_ms_stats` (min / median / p95 over >= 200 further steps),
`kernels` (per-kernel roofline from event pairs on the launch stream), `frame` (MEASURED tracking+mapping ms/frame of a
31-frame sequence with the reference's host sampling run ahead by producer threads, and with device sampling),
`roofline`, `cpu_baseline`.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import random
import sys
import time



def usable_cores():
    """CPU threads this process may really use: min(affinity mask, cgroup CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# one NUMA node per rank (mipsfusion_amd/hostcpu.py: the host sample producers run 9 or 14 ms per frame otherwise)
# (loaded by path: importing the package would import torch before the OpenMP variables below are set)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("mipsf_hostcpu", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                           "mipsfusion_amd", "hostcpu.py"))
_hostcpu = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_hostcpu)


def _only_launches_ranks(argv):
    """True for `python bench.py --gpus N` (N > 1) without a launcher: this process only starts the ranks (launch_ranks) and
    must leave its affinity mask alone -- the children inherit it and each picks its OWN NUMA node from the full mask."""
    if "WORLD_SIZE" in os.environ:
        return False
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return argv[i + 1].isdigit() and int(argv[i + 1]) > 1
        if a.startswith("--gpus="):
            return a[7:].isdigit() and int(a[7:]) > 1
    return False


HOST_CPUS = None if (os.environ.get("MIPSF_NO_CONFINE") or _only_launches_ranks(sys.argv[1:])) else _hostcpu.confine_to_numa_node(
    32, int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))

# OpenMP sizes its pools by the machine's core count (256 on the GPU hosts) although the cgroup grants 16: every host
# thread that touches a torch CPU op would oversubscribe the quota.  Must be set before torch is imported.
os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mipsfusion_amd import dist as mdist  # noqa: E402
from mipsfusion_amd import ops, synth  # noqa: E402
from mipsfusion_amd.helper_functions import sampling_helper as sh  # noqa: E402
from mipsfusion_amd.helper_functions.geometry_helper import matrix_to_quaternion, qt_to_transform_matrix  # noqa: E402
from mipsfusion_amd.helper_functions.utils import backward_from_one, get_loss_from_ret  # noqa: E402
from mipsfusion_amd.graph import GraphedSteps, work_stream  # noqa: E402
from mipsfusion_amd.model import JointEncoding  # noqa: E402
from mipsfusion_amd.optim import FusedAdam  # noqa: E402

N_RAYS, N_SAMPLES = 4096, 64
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: f16 / bf16 MFMA dense peak
N_GRID_PARAMS, N_DEC_PARAMS = 9014144, 36577

# algorithmic cost per unit (SURVEY.md 8d / BASELINE.md 3); unit = 1 ray*sample.
# (bound, bytes per LIVE unit, bytes per DEAD unit, liveness that applies, arithmetic type, matrix products issued per
#  algorithmic product or None).  A sample behind the truncation band has an exactly zero gradient (DESIGN_NOTES 4e): the
# backward chain short-cuts its 32-sample tile, the weight-gradient and dx kernels never visit it, the scatter makes no
# record for a (sample, level) pair with a zero feature gradient.  `work` of a launch = the units it PROCESSED:
#   liveness "tile": share of live 32-sample tiles of the step (the chain kernel's own lists), "pair": share of live
#   (sample, level) pairs, None: every unit.  The full-batch figure (every unit priced as live) is kept as `frac_full_batch`.
# The two f16x3 decoder kernels keep or leave the activation record of the backward pass: 1.7-1.8 KB of HBM traffic per
# sample against 0.22 MFLOP at 2.5 PFLOP/s -- they are priced against the HBM roofline, their matrix-pipe utilisation is
# reported next to it.
_F16X3 = "f16x3 (f16 MFMA on hi/lo split operands = 22-23 operand bits, fp32 accumulate)"
_BF16X6 = "bf16x6 (fp32 operands carried exactly as three bf16 pieces, six bf16 MFMAs per product, fp32 accumulate)"
_F32 = "f32 (fp32-input MFMA = exact fp32 products)"


def kernel_cost(precision):
    """name -> (bound, bytes per live unit, bytes per dead unit, liveness, dtype, matrix products issued per algorithmic one,
    design bytes per unit on top of the algorithmic ones).  The decoder rows follow the arithmetic's records:
      bf16x6 (default), f16x3 (fast mode)   lean records: H1 / dG3 / the rgb_emb half of dH2 are not written, the exchange
                        form of the streaming weight-gradient kernel recomputes them
      f32               the round-1 fp32-MFMA kernels: full records, no zero-tile short cut."""
    cost = {
        # 8 corners x 16 levels x 2 features gathered (1024) + coordinates (12) + features out (128); the
        # Jacobian the design also writes for the backward (384 B) is reported as design_bytes, not priced
        "hashgrid_fwd": ("hbm", 1164.0, 0.0, None, "f32", None, 384.0),
        # scatter: x + dL/dy + read-modify-write of the touched entries; a dead pair costs the 8-byte read that finds it dead
        "hashgrid_bwd": ("hbm", 2188.0, 128.0, "pair", "f32+f64 LDS", None, 0.0),
        "hashgrid_dx": ("hbm", 536.0, 0.0, "tile", "f32", None, 0.0),   # saved Jacobian (384) + dL/dy (128) + dx read-modify-write (24)
        "sample_rays": ("hbm", 20.0, 0.0, None, "f32+f64", None, 0.0),
        "render_fwd": ("hbm", 44.0, 0.0, None, "f32", None, 0.0),
        "render_bwd": ("hbm", 84.0, 0.0, None, "f32", None, 0.0),
        "rays_bwd": ("hbm", 16.0, 0.0, None, "f32+f64", None, 0.0),
    }
    io_f, io_b, dead_b = 12.0 + 128.0 + 40.0, 40.0 + 40.0 + 32.0 + 12.0 + 128.0 + 12.0, 40.0 + 128.0 + 12.0
    if precision == "f16x3":
        cost["decoder_fwd"] = ("hbm", io_f + 1024.0 + 32.0, 0.0, None, _F16X3, 3, 0.0)             # H2 + H3 + masks
        cost["decoder_bwd_chain"] = ("hbm", io_b + 768.0 + 32.0, dead_b, "tile", _F16X3, 3, 0.0)     # dG1 + sdf_emb half of dH2 + small rows
        cost["decoder_wgrad"] = ("hbm", 1024.0 + 768.0 + 32.0 + 16.0 + 128.0 + 12.0, 0.0, "tile",
                                 _F16X3 + ", gradient blocks under per-block power-of-two scales", 3, 0.0)
    elif precision == "bf16x6":
        # SURVEY 8(d) prices the decoder in FLOPs against the MFMA roofline.  In this arithmetic an fp32-exact product IS six
        # bf16 MFMAs, and the kernels are bound by instruction issue (PMC, profiles/r04_*: matrix pipe 36-50 % busy, the vector
        # ALU as long again, the two barely overlap within a SIMD), not by their lean records (2.0-2.4 TB/s): bound "mfma",
        # achieved = ISSUED matrix FLOP/s of the units processed against the dense 16-bit peak; the algorithmic figure and the
        # HBM view of the same launch sit next to it (matrix_pipe / hbm_view).
        cost["decoder_fwd"] = ("mfma", DECODER_FLOP_PER_SAMPLE, 0.0, None, _BF16X6, 6, 0.0, io_f + 1024.0 + 32.0, 0.0)
        cost["decoder_bwd_chain"] = ("mfma", DECODER_FLOP_PER_SAMPLE, 0.0, "tile", _BF16X6, 6, 0.0, io_b + 768.0 + 32.0, dead_b)
        cost["decoder_wgrad"] = ("mfma", DECODER_FLOP_PER_SAMPLE, 0.0, "tile", _BF16X6, 6, 0.0,
                                 1024.0 + 768.0 + 32.0 + 16.0 + 128.0 + 12.0, 0.0)
    else:
        cost["decoder_fwd"] = ("mfma", DECODER_FLOP_PER_SAMPLE, 0.0, None, _F32, 1, 0.0)
        cost["decoder_bwd_chain"] = ("mfma", DECODER_FLOP_PER_SAMPLE, 0.0, None, _F32, 1, 0.0)
        cost["decoder_wgrad"] = ("mfma", DECODER_FLOP_PER_SAMPLE, 0.0, None, _F32, 1, 0.0)
    return cost


DECODER_FLOP_PER_SAMPLE = 72370.0
# device streaming rates measured with hand-written kernels (tools/micro/stream.hip, 1 GiB, 16 B per lane, >= 32 KB in
# flight per CU; profiles/r03_stream.txt): what "the HBM roofline" is worth on this part for each access mix
STREAM_CEILINGS_TBS = {"read": 6.3, "read_nt": 7.0, "write": 6.0, "copy": 5.6, "spec": 8.0}


DTYPE_OF = {        # <= 120 characters (the long description of each arithmetic: DESIGN.md 2)
    "bf16x6": "f32; decoder products as 3 x bf16 pieces of each fp32 operand, 6 bf16 MFMAs per product, fp32 accumulate",
    "f16x3": "f32; decoder products as f16 hi/lo halves (22-23 operand bits), 3 f16 MFMAs per product, fp32 accumulate",
    "f32": "f32 everywhere (fp32-input MFMA = exact fp32 products)",
}
_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


class BoardSampler:
    """The board's state while a region is timed: the shader clock the SMU reports (sysfs pp_dpm_sclk, the level marked '*') and
    the package power (hwmon power1_average / power1_input), sampled every `period` seconds from a second thread.  The decoder
    kernels hold the board at its power cap and the clock it then delivers differs from box to box by more than most kernel
    changes: a step time is only comparable with the board state it was measured at."""

    def __init__(self, device_index=0, period=0.004):
        import glob
        import threading
        self.period, self.samples, self._stop, self._glob = period, [], False, glob
        cards = [d for d in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
        mine = []
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            mine = [d for d in cards if addr in os.path.realpath(d)]
        except Exception:       # noqa: BLE001
            pass
        self.card = (mine or cards or [None])[0]
        self._power_files = []
        if self.card:
            self._power_files = (glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "power1_average")) +
                                 glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "power1_input")))
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _sclk(self):
        try:
            for line in open(os.path.join(self.card, "pp_dpm_sclk")):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError, TypeError):
            pass
        return None

    def _power(self):
        for f in self._power_files:
            try:
                return int(open(f).read()) * 1e-6
            except (OSError, ValueError):
                pass
        return None

    def _run(self):
        while not self._stop:
            self.samples.append((self._sclk(), self._power()))
            time.sleep(self.period)

    def __enter__(self):
        if self.card:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self.card:
            self._thread.join(timeout=2.0)

    def summary(self):
        def med(v):
            v = sorted(x for x in v if x is not None)
            return None if not v else round(float(v[len(v) // 2]), 1)
        clk, pw = [a for a, _ in self.samples], [b for _, b in self.samples]
        lo = [x for x in clk if x is not None]
        return {"sclk_mhz_median": med(clk), "sclk_mhz_min": round(min(lo), 1) if lo else None, "power_w_median": med(pw),
                "samples": len(self.samples), "source": "sysfs pp_dpm_sclk / hwmon power1_*, sampled across the graph-replayed "
                "timed region and the step_ms_stats steps" if self.card else "no sysfs card with pp_dpm_sclk visible"}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--setup-iters", type=int, default=50, help="untimed mapping iterations so the SDF has sign changes")
    p.add_argument("--cpu-rays", type=int, default=4096, help="rays of the bounded CPU-baseline sample (0 = skip)")
    p.add_argument("--cpu-iters", type=int, default=10)
    p.add_argument("--cpu-warmup", type=int, default=3)
    p.add_argument("--no-frame-estimate", action="store_true")
    p.add_argument("--stats-steps", type=int, default=200, help="further steps timed one replay at a time (min/median/p95)")
    p.add_argument("--seq-frames", type=int, default=31, help="frames of the measured tracking+mapping sequence (0 = skip)")
    p.add_argument("--config3-frames", type=int, default=300, help="frames of the multi-sub-map sequence (0 = skip)")
    p.add_argument("--no-graph", action="store_true",
                   help="time eager launches instead of hipGraph replays of pose_accum_step iterations")
    p.add_argument("--no-variants", action="store_true", help="skip the f32 / dense / unchanged-caller steps")
    p.add_argument("--torch-pose", action="store_true",
                   help="build rays with the reference's eager torch ops (mipsfusion.py:320-322) instead of the fused op")
    return p.parse_args()


def build_submap(cfg, dev, seed):
    """One submap's state: model, 4 keyframes + current frame, optimisable keyframe poses, host-sampled ray pool."""
    random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)
    bb = torch.from_numpy(np.array(cfg["mapping"]["bound"]))
    nf = torch.from_numpy(np.array(cfg["mapping"]["localMLP_max_len"]))
    model = JointEncoding(cfg, bb, nf)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        model.embed_fn.params.copy_((torch.rand(model.embed_fn.params.shape, generator=g) * 2 - 1) * 1e-4)
    model = model.to(dev).train()
    model.accumulate_param_grads_in_place = True      # plain loss.backward() loop: opt in (scene_rep._QueryFn)
    # one backward per map step and map_opt.step(zero_grad=True) clears the gradients: only then may the scatter store
    model.grid_grad_is_zero_at_backward = cfg["mapping"]["map_accum_step"] == 1 and cfg["mapping"].get("map_wait_step", 0) == 0
    # keyframes on a small arc + the current frame
    frames, poses = [], []
    for k in range(5):
        c2w = synth.default_pose(cfg, yaw=0.3 + 0.12 * k, pitch=-0.1 + 0.02 * k)
        c2w[:3, 3] += torch.tensor([0.05 * k, 0.08 * k, 0.0])
        frames.append(synth.make_frame(cfg, c2w, seed=seed * 100 + k, frame_id=k))
        poses.append(c2w)
    poses = torch.stack(poses)
    return model, frames, poses


def build_ray_table(cfg, frames, dev):
    """[4 keyframes x R rows | H*W rows of the current frame] in ONE device table (DeviceRayDB storage)."""
    from mipsfusion_amd.keyframe_rays import DeviceRayDB
    H, W = frames[0]["depth"].shape
    rows, cols = sh.sample_pixels_uniformly(H, W, 100, 300)              # kf_n_rays_h x kf_n_rays_w (scaled: 30 000)
    R, n_kf = rows.shape[0], len(frames) - 1
    table = torch.zeros(n_kf * R + H * W, 7, device=dev)
    db = DeviceRayDB(n_kf, R, dev, storage=table)
    for k in range(n_kf):
        f = frames[k]
        db.store(k, torch.cat([f["direction"], f["rgb"], f["depth"][..., None]], -1)[rows, cols].to(dev))
    cur = frames[-1]
    table[n_kf * R:].copy_(torch.cat([cur["direction"], cur["rgb"], cur["depth"][..., None]], -1).reshape(-1, 7))
    return table, db, R


def draw_index_sets(cfg, frames, db, R, n_sets):
    """Host pixel sampling exactly as local_BA does it (mipsfusion.py:295-317): keyframe rays by the python
    random.sample calls of sample_rays_in_submap, current-frame pixels by sample_pixels_mix -> [n_sets, N] table rows
    and the index of the owning pose."""
    H, W = frames[0]["depth"].shape
    n_kf = len(frames) - 1
    n_cur, n_from_kf = cfg["mapping"]["pixels_cur"], N_RAYS - cfg["mapping"]["pixels_cur"]
    related = torch.arange(n_kf)
    rows_l, own_l = [], []
    for _ in range(n_sets):
        flat, _, kf_indices = db.indices_in_submap(related[0], related, n_from_kf)
        r, c = sh.sample_pixels_mix(H, W, cfg["tracking"]["RO"]["n_rows"], cfg["tracking"]["RO"]["n_cols"],
                                    frames[-1]["depth"], n_cur)
        rows_l.append(torch.cat([flat, n_kf * R + r * W + c]))
        own_l.append(torch.cat([kf_indices, torch.full((n_cur,), n_kf, dtype=torch.int64)]))
    return torch.stack(rows_l), torch.stack(own_l)


class MappingLoop:
    """The local-BA iteration of mipsfusion.py:293-342 against our JointEncoding."""

    def __init__(self, cfg, model, poses, table, idx_rows, idx_owner, dev, torch_pose=False, capturable=False):
        self.cfg, self.model, self.dev, self.torch_pose = cfg, model, dev, torch_pose
        self.map_opt = FusedAdam([{"params": model.decoder.parameters(), "weight_decay": 1e-6, "lr": cfg["mapping"]["lr_decoder"]},
                                  {"params": model.embed_fn.parameters(), "eps": 1e-15, "lr": cfg["mapping"]["lr_embed"]}],
                                 betas=(0.9, 0.99), capturable=capturable)
        poses = poses.to(dev)
        self.pose_fixed = poses[:1]                                  # first keyframe stays fixed
        self.cur_trans = torch.nn.Parameter(poses[1:, :3, 3].clone())
        self.cur_rot = torch.nn.Parameter(matrix_to_quaternion(poses[1:, :3, :3]))
        # pose Adam: same fused kernel as the map (dense torch.optim.Adam semantics, tests/test_gpu_parity.py); the
        # torch optimiser costs ~12 tiny launches per pose step
        self.pose_opt = FusedAdam([{"params": self.cur_rot, "lr": cfg["mapping"]["lr_rot"]},
                                   {"params": self.cur_trans, "lr": cfg["mapping"]["lr_trans"]}],
                                  capturable=capturable)
        self.poses_all = torch.cat([self.pose_fixed, qt_to_transform_matrix(self.cur_rot, self.cur_trans)], 0)
        self.table = table
        self.idx_rows, self.idx_owner = idx_rows.to(dev), idx_owner.to(dev)          # [n_sets, N]: one FRESH batch per step
        self.n_sets = self.idx_rows.shape[0]
        self.noise = torch.rand(self.n_sets, N_RAYS, N_SAMPLES, device=dev)
        self.i = 0

    def make_static(self, n_inner):
        """Static input buffers (table rows, owner, jitter) of a captured group of n_inner iterations."""
        self.stk = [self.idx_rows, self.idx_owner, self.noise]
        self.static = [t[:n_inner].clone() for t in self.stk]
        self.n_inner = n_inner
        self._arange = torch.arange(n_inner, device=self.dev)

    def refill_static(self):
        """the next n_inner UNUSED batches -> the graph's static inputs (three device-to-device copies)"""
        idx = (self._arange + self.i) % self.n_sets
        for dst, src in zip(self.static, self.stk):
            torch.index_select(src, 0, idx, out=dst)

    def step_static(self, k):
        rows, owner, noise = (t[k] for t in self.static)
        return self._iterate(rows, owner, noise, set_to_none=False)

    def step(self):
        b = self.i % self.n_sets
        return self._iterate(self.idx_rows[b], self.idx_owner[b], self.noise[b], set_to_none=True)

    def batch(self, b):
        """(rays [N,7], owner) of index set b, for the forward-only / CPU legs"""
        return ops.gather_rays(self.table, self.idx_rows[b]), self.idx_owner[b]

    def _iterate(self, rows, owner, noise, set_to_none):
        cfg = self.cfg
        if self.torch_pose:
            rays_d_cam, target_s, target_d = ops.gather_rays(self.table, rows, split=True)  # fresh rows of the ray table
            rays_d = torch.sum(rays_d_cam[..., None, :] * self.poses_all[owner, :3, :3], -1)
            rays_o = self.poses_all[owner, :3, -1]
            ret = self.model.forward(rays_o, rays_d, target_s, target_d, noise=noise)
        else:   # fresh rows of the ray table, the same pose arithmetic and the sample placement in one kernel (the pose
            #     gradients go straight to .grad, from one kernel as well)
            ret = self.model.forward_from_table(self.table, rows, self.cur_rot, self.cur_trans, self.pose_fixed, owner, noise,
                                                accumulate_in_place=True)
        loss = get_loss_from_ret(ret, cfg["training"])
        backward_from_one(loss, retain_graph=self.torch_pose)     # = loss.backward(), minus autograd's ones_like fill
        self.i += 1
        if self.i % cfg["mapping"]["map_accum_step"] == 0:
            self.map_opt.step(zero_grad=True)           # step + zero_grad (mipsfusion.py:330-335) in one pass
        if self.i % cfg["mapping"]["pose_accum_step"] == 0:
            self.pose_opt.step()
            if self.torch_pose:
                self.poses_all = torch.cat([self.pose_fixed, qt_to_transform_matrix(self.cur_rot, self.cur_trans)], 0)
            self.pose_opt.zero_grad(set_to_none=set_to_none)
        return loss


class UnchangedCallerLoop:
    """What the three import lines of INTEGRATION.md section 1 alone deliver: the reference's own local-BA iteration
    (mipsfusion.py:293-342) as the unchanged caller runs it -- host pixel sampling per iteration (keyframe rays by python
    random.sample, current frame by sample_pixels_mix), CPU gather of the ray rows + three .to(device) copies, eager torch
    ray ops (:320-322), JointEncoding.forward drawing its jitter with torch.rand on the CPU + upload (scene_rep.py:176),
    get_loss_from_ret's selects, loss.backward(retain_graph=True) with autograd-path parameter gradients, torch.optim.Adam
    for map and poses (:580-584, :300-303), zero_grad() calls -- no fused ray op, no FusedAdam, no in-place gradient
    accumulation, no pre-drawn indices, no hipGraph.  Runs on a deep copy of the submap."""

    def __init__(self, cfg, model, frames, poses, table, db, R, dev):
        import copy
        self.cfg, self.dev, self.frames, self.db, self.R = cfg, dev, frames, db, R
        self.model = copy.deepcopy(model)
        self.model.accumulate_param_grads_in_place = False
        self.table_cpu = table.cpu()
        self.map_opt = torch.optim.Adam([{"params": self.model.decoder.parameters(), "weight_decay": 1e-6, "lr": cfg["mapping"]["lr_decoder"]},
                                         {"params": self.model.embed_fn.parameters(), "eps": 1e-15, "lr": cfg["mapping"]["lr_embed"]}],
                                        betas=(0.9, 0.99))
        poses = poses.to(dev)
        self.pose_fixed = poses[:1]
        self.cur_trans = torch.nn.Parameter(poses[1:, :3, 3].clone())
        self.cur_rot = torch.nn.Parameter(matrix_to_quaternion(poses[1:, :3, :3]))
        self.pose_opt = torch.optim.Adam([{"params": self.cur_rot, "lr": cfg["mapping"]["lr_rot"]},
                                          {"params": self.cur_trans, "lr": cfg["mapping"]["lr_trans"]}])
        self.poses_all = torch.cat([self.pose_fixed, qt_to_transform_matrix(self.cur_rot, self.cur_trans)], 0)
        self.i = 0

    def iterate(self, predrawn=None, marks=None, detach_rays=False):
        """marks (a dict): wall-clock of every section, each closed by a device synchronisation (the breakdown pass only);
        detach_rays: the poses get no gradient (prices the backward of the caller's own eager ray ops, mipsfusion.py:320-322)"""
        cfg, dev = self.cfg, self.dev
        t_prev = [time.perf_counter()]

        def mark(name):
            if marks is not None:
                torch.cuda.synchronize()
                now = time.perf_counter()
                marks[name] = marks.get(name, 0.0) + (now - t_prev[0]) * 1e3
                t_prev[0] = now
        if predrawn is None:
            rows, owner = draw_index_sets(cfg, self.frames, self.db, self.R, 1)
            rows, owner = rows[0], owner[0]
        else:
            rows, owner = predrawn
        mark("host_pixel_sampling")
        rays = self.table_cpu[rows]                                         # CPU gather (keyframeSet.py:386-436)
        rays_d_cam, target_s, target_d = rays[..., :3].to(dev), rays[..., 3:6].to(dev), rays[..., 6:7].to(dev)
        owner = owner.to(dev)
        mark("cpu_ray_gather_and_uploads")
        rays_d = torch.sum(rays_d_cam[..., None, :] * self.poses_all[owner, :3, :3], -1)
        rays_o = self.poses_all[owner, :3, -1]
        if detach_rays:
            rays_d, rays_o = rays_d.detach(), rays_o.detach()
        mark("eager_torch_ray_ops")
        ret = self.model.forward(rays_o, rays_d, target_s, target_d)        # jitter: the CPU generator's draw + upload
        ret = {k: v for k, v in ret.items() if not k.startswith("_")}       # the reference's dictionary keys only
        loss = get_loss_from_ret(ret, cfg["training"])
        mark("forward_incl_cpu_jitter_draw_and_upload")
        loss.backward(retain_graph=True)
        mark("backward_autograd_path_gradients")
        self.i += 1
        if self.i % cfg["mapping"]["map_accum_step"] == 0:
            self.map_opt.step()
            self.map_opt.zero_grad()
        mark("torch_optim_adam_map_step_and_zero_grad")
        if self.i % cfg["mapping"]["pose_accum_step"] == 0:
            self.pose_opt.step()
            self.poses_all = torch.cat([self.pose_fixed, qt_to_transform_matrix(self.cur_rot, self.cur_trans)], 0)
            self.pose_opt.zero_grad()
        mark("pose_step_and_pose_matrices")
        return loss


def unchanged_caller_device_split(loop, pre_rows, pre_owner, n=10):
    """Whose kernels the unchanged caller's iteration runs: a torch.profiler pass over n iterations (indices pre-drawn), device
    time and launches per iteration of the drop-in modules' kernels (mipsf::*) and of everything else -- the caller's own eager
    torch ops and their autograd (the poses_all[indices] gathers' backward is one indexing_backward kernel each), torch.optim.Adam,
    the uploads."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for k in range(n):
            loop.iterate((pre_rows[k % 8], pre_owner[k % 8]))
        torch.cuda.synchronize()
    split = {"modules_kernels": [0.0, 0], "torch_kernels_of_the_caller_and_torch_optim": [0.0, 0], "copies_and_fills_by_the_runtime": [0.0, 0]}
    top = []
    for e in prof.key_averages():
        if e.device_type != DeviceType.CUDA or not ("(" in e.key or e.key.startswith("Mem")):       # kernels and copies, not ranges
            continue
        us = getattr(e, "self_device_time_total", None)
        us = e.self_cuda_time_total if us is None else us
        if "mipsf::" in e.key:
            k = "modules_kernels"
        elif e.key.startswith("Memcpy") or e.key.startswith("Memset"):
            k = "copies_and_fills_by_the_runtime"
        else:
            k = "torch_kernels_of_the_caller_and_torch_optim"
            top.append((us / n * 1e-3, e.count / n, e.key))
        split[k][0] += us
        split[k][1] += e.count
    out = {k: {"ms_per_step": round(v[0] / n * 1e-3, 4), "launches_per_step": round(v[1] / n, 1)} for k, v in split.items()}
    out["largest_torch_kernels"] = [{"ms_per_step": round(a, 4), "launches_per_step": round(b, 1), "kernel": c[:90]}
                                    for a, b, c in sorted(top, reverse=True)[:4]]
    out["note"] = ("device time, not wall clock: the step is what these add up to plus the launch gaps of ~200 eager launches; the "
                   "modules' share is what a caller-side change cannot touch, the rest is the reference's own torch code")
    return out


def unchanged_caller_rate(cfg, model, frames, poses, table, db, R, dev, steps):
    loop = UnchangedCallerLoop(cfg, model, frames, poses, table, db, R, dev)
    pre_rows, pre_owner = draw_index_sets(cfg, frames, db, R, 8)
    out = {}
    for key, pre in (("ms_per_step", False), ("ms_per_step_indices_predrawn", True)):
        for k in range(3):
            loop.iterate((pre_rows[k % 8], pre_owner[k % 8]) if pre else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            loop.iterate((pre_rows[k % 8], pre_owner[k % 8]) if pre else None)
        torch.cuda.synchronize()
        out[key] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    # where the time goes: the same iteration with a device synchronisation behind every section (so the sections add up to
    # MORE than the pipelined step above: host work no longer overlaps the GPU's)
    marks, n_b = {}, 10
    for k in range(n_b):
        loop.iterate(None, marks)
    out["breakdown_ms_per_step_serialised"] = {k: round(v / n_b, 4) for k, v in marks.items()}
    marks2 = {}
    for k in range(n_b):
        loop.iterate(None, marks2, detach_rays=True)
    out["breakdown_ms_per_step_serialised"]["(of backward) without the caller's eager ray ops in the graph"] = round(
        marks2["backward_autograd_path_gradients"] / n_b, 4)
    out["breakdown_note"] = ("the backward section contains the autograd of the CALLER's own eager torch ray ops (poses_all[indices] gather -> "
                             "index_put with a sort, the quaternion chain: ~150 small launches) next to the modules' six kernels; the "
                             "line above it without them is what the drop-in modules themselves cost")
    out["device_time_by_owner"] = unchanged_caller_device_split(loop, pre_rows, pre_owner)
    t0 = time.perf_counter()
    for _ in range(10):
        torch.rand(N_RAYS, N_SAMPLES)
    out["breakdown_ms_per_step_serialised"]["(of forward) cpu_jitter_draw_alone"] = round((time.perf_counter() - t0) / 10 * 1e3, 4)
    out["value"] = round(N_RAYS * N_SAMPLES / (out["ms_per_step"] * 1e-3), 1)
    out["unit"] = "rays*samples/s"
    out["what"] = ("the reference's own local_BA iteration (mipsfusion.py:293-342) over the drop-in modules and nothing "
                   "else: host pixel sampling per iteration, CPU ray gather + uploads, eager torch ray ops, CPU torch.rand "
                   "jitter + upload, autograd-path gradients, torch.optim.Adam, eager launches")
    del loop
    torch.cuda.empty_cache()
    return out


def graphed_ms_per_step(loop, stream, n_inner, steps, warmup):
    """ms per step of `steps` graph-replayed mapping steps with the model / ops switches as they are now."""
    loop.refill_static()
    g = GraphedSteps(loop.step_static, n_inner, stream=stream)
    for _ in range(max(1, warmup // n_inner)):
        loop.refill_static()
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // n_inner):
        loop.refill_static()
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    del g
    return ms


def forward_only_rate(model, loop, dev, iters=10):
    rays, owner = loop.batch(0)
    poses_now = torch.cat([loop.pose_fixed, qt_to_transform_matrix(loop.cur_rot, loop.cur_trans)], 0).detach()
    with torch.no_grad():
        rays_d = torch.sum(rays[:, :3][..., None, :] * poses_now[owner, :3, :3], -1).contiguous()
        rays_o = poses_now[owner, :3, -1].contiguous()
        model.eval()
        for _ in range(3):
            model.forward(rays_o, rays_d, None, rays[:, 6:7], noise=loop.noise[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            model.forward(rays_o, rays_d, None, rays[:, 6:7], noise=loop.noise[0])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    model.train()
    return N_RAYS * N_SAMPLES / dt, dt * 1e3


def decoder_gemm_rates(model, dev, M):
    """The decoder's batched (rays*samples x feat) x (feat x hidden) products on their own: the forward kernel WITHOUT the
    activation record of the backward pass (the evaluation branch / RandomOptimizer rounds), per arithmetic, with the
    matrix-pipe utilisation = matrix-core FLOP issued / dense 16-bit peak (bf16x6 issues six products per algorithmic one,
    f16x3 three)."""
    from mipsfusion_amd._lib import FEAT_LEVEL_MAJOR
    ws = model.decoder.ordered_parameters()
    packs = {"f16x3": ops.decoder_pack16(ws), "bf16x6": ops.decoder_pack16(ws, precision="bf16x6")}
    packs["f16"] = packs["f16x3"]
    x = torch.rand(M, 3, device=dev)
    feat = torch.randn(16, M, 2, device=dev) * 1e-2
    out = {}
    for prec, mult in (("bf16x6", 6), ("f16x3", 3), ("f16", 1)):
        fn = lambda: ops.decoder_fwd(None, feat, FEAT_LEVEL_MAJOR, x, None, M, save=False, precision=prec, packed16=packs[prec])   # noqa: E731
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        issued = DECODER_FLOP_PER_SAMPLE * M * mult / (ms * 1e-3) / 1e12
        out[prec] = {"ms": round(ms, 4), "algorithmic_tflops": round(issued / mult, 2), "issued_tflops": round(issued, 2),
                     "peak_tflops": MFMA_F16_PEAK_TFLOPS, "utilisation": round(issued / MFMA_F16_PEAK_TFLOPS, 4)}
    out["note"] = ("forward kernel without the activation record, 262 144 samples; the training forward (kernels.decoder_fwd) also "
                   "writes 1 KB of record per sample and is priced against HBM")
    return out


def inference_rates(cfg, model, dev):
    """Logger.render_full_img-style full-frame render (10 000-ray chunks) and Mesher-style dense grid
    queries (16 384-point batches) through the eval / query_* entry points."""
    from mipsfusion_amd import inference
    frame = synth.make_frame(cfg, seed=2)
    H, W = frame["depth"].shape
    S = cfg["training"]["n_samples_d"] + cfg["training"]["n_range_d"]
    noise = torch.rand(H * W, S, device=dev)
    was_training = model.training
    model.eval()
    d_cam, c2w, depth = frame["direction"].to(dev), frame["c2w"].to(dev), frame["depth"].to(dev)

    def render():
        return inference.render_full_img(model, d_cam, c2w, depth, H, W, 10000, noise=noise)
    render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        render()
    torch.cuda.synchronize()
    img_ms = (time.perf_counter() - t0) / 3 * 1e3
    pts = torch.rand(128 ** 3, 3, device=dev)
    inference.query_in_batches(model.query_sdf, pts[:65536])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inference.query_in_batches(model.query_sdf, pts, batch_size=1024 * 16)
    torch.cuda.synchronize()
    q16k_ms = (time.perf_counter() - t0) * 1e3
    inference.query_in_batches(model.query_sdf, pts, batch_size=1024 * 1024)      # warm the allocator at this batch size
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inference.query_in_batches(model.query_sdf, pts, batch_size=1024 * 1024)
    torch.cuda.synchronize()
    q1m_ms = (time.perf_counter() - t0) * 1e3
    model.train(was_training)
    return {"full_image_ms": round(img_ms, 3), "image": f"{W}x{H} rays x {S} samples, 10000-ray chunks",
            "full_image_rays_samples_per_s": round(H * W * S / (img_ms * 1e-3), 1),
            "grid_query_128cubed_ms_batch16k": round(q16k_ms, 3), "grid_query_128cubed_ms_batch1M": round(q1m_ms, 3),
            "grid_query_points_per_s_batch1M": round(128 ** 3 / (q1m_ms * 1e-3), 1)}


def frame_estimate(cfg, model, loop, dev, ba_ms, stream=None):
    """tracking+mapping ms/frame = iter_RO*RO + tracking.iter*GO + mapping.iters*BA/map_every."""
    # RO: the whole RandomOptimizer.optimize call of one frame (iter_RO fused rounds: particles -> grid -> decoder ->
    # fitness -> swarm update, search state on the device, one read-back at the end), divided by iter_RO
    import types
    from mipsfusion_amd.RandomOptimizer import RandomOptimizer
    rcfg = cfg["tracking"]["RO"]
    rcfg.setdefault("initial_scaling_factor", 0.02)
    rcfg.setdefault("rescaling_factor", 0.5)
    cfg["tracking"].setdefault("ignore_edge_W", 20)
    cfg["tracking"].setdefault("ignore_edge_H", 20)
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    frame = synth.make_frame(cfg, seed=1)
    ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frame["direction"])
    ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
    n_ro = max(1, cfg["tracking"]["iter_RO"])
    init = frame["c2w"].clone()
    was_training = model.training
    model.eval()
    ro_ms_by = {}
    for prec in (model.decoder_precision, "f16x3", "f16"):   # the default arithmetic; the two opt-in fast ones next to it
        ro.decoder_precision = prec                          # ("f16": BASELINE config 5 "fp16 decoder on CDNA4", pose within 1e-3)
        for _ in range(2):
            ro.optimize(model, frame["depth"], init, None, n_iter=n_ro)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ro.optimize(model, frame["depth"], init, None, n_iter=n_ro)
        torch.cuda.synchronize()
        ro_ms_by[prec] = (time.perf_counter() - t0) / 5 / n_ro * 1e3
    ro_ms = ro_ms_by[model.decoder_precision]
    model.train(was_training)
    # GO: tracking.sample rays, pose-only Adam on one pose
    ns = cfg["tracking"]["sample"]
    rays = loop.batch(0)[0][:ns]
    rot = torch.nn.Parameter(loop.cur_rot.detach()[-1:].clone())
    trans = torch.nn.Parameter(loop.cur_trans.detach()[-1:].clone())
    popt = FusedAdam([{"params": rot, "lr": 1e-3}, {"params": trans, "lr": 1e-3}], capturable=stream is not None)
    noise = loop.noise[0][:ns]

    d_cam, t_rgb, t_d = rays[:, :3].contiguous(), rays[:, 3:6].contiguous(), rays[:, 6:7].contiguous()
    own = torch.zeros(ns, dtype=torch.int64, device=dev)
    rows_go = loop.idx_rows[0][:ns].contiguous()

    def go():
        if loop.torch_pose:
            c2w = qt_to_transform_matrix(rot, trans)
            rays_o = c2w[..., :3, -1].repeat(ns, 1)
            rays_d = torch.sum(d_cam[..., None, :] * c2w[:, :3, :3], -1)
            ret = model.forward(rays_o, rays_d, t_rgb, t_d, EMD_w=0., noise=noise)
        else:       # the tracking iteration of the graphed sequences: rows of the ray table, one launch each way for the rays
            ret = model.forward_from_table(loop.table, rows_go, rot, trans, None, own, noise, EMD_w=0., accumulate_in_place=True)
        get_loss_from_ret(ret, cfg["training"]).backward()
        popt.step(zero_grad=True)       # (the pose gradients are cleared by the optimiser kernel itself)
    def time_go():
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            go()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 10 * 1e3
    zero_promise, model.grid_grad_is_zero_at_backward = model.grid_grad_is_zero_at_backward, False    # (backward passes pile up here)
    go_ms_unfrozen = time_go()          # reference as shipped: freeze_model() is a no-op (typo `require_grad`)
    for prm in model.parameters():      # what freeze_model (mipsfusion.py:226-230) intends; pose results identical
        prm.requires_grad_(False)
    go_ms = time_go()
    go_graph_ms = None
    if stream is not None and not loop.torch_pose:      # a frame's tracking iterations as one hipGraph replay, the map
        n_go = max(1, cfg["tracking"]["iter"])           # frozen throughout: the operand images are packed once per replay

        def go_k(k):                                     # (GraphedSequence._go_step: the pack is recorded at k == 0)
            if k == 0:
                model.frozen_weights(True)
            go()
            if k == n_go - 1:
                model.frozen_weights(False)
        g = GraphedSteps(go_k, n_go, stream=stream)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        go_graph_ms = (time.perf_counter() - t0) / 20 / n_go * 1e3
    for prm in model.parameters():
        prm.requires_grad_(True)
    model.zero_grad()
    model.grid_grad_is_zero_at_backward = zero_promise
    go_eager_ms = go_ms
    if go_graph_ms is not None:
        go_ms = go_graph_ms
    tr, mp = cfg["tracking"], cfg["mapping"]
    total = tr["iter_RO"] * ro_ms + tr["iter"] * go_ms + mp["iters"] * ba_ms / mp["map_every"]
    return {"ro_iter_ms": round(ro_ms, 4), "ro_decoder_arithmetic": f"{model.decoder_precision} (the RandomOptimizer's default)",
            "ro_iter_ms_by_arithmetic": {k: round(v, 4) for k, v in ro_ms_by.items()},
            "decoder_arithmetic": model.decoder_precision,
            "go_iter_ms": round(go_ms, 4),
            "go_iter_ms_eager": round(go_eager_ms, 4),
            "go_iter_ms_map_grads_computed_and_discarded": round(go_ms_unfrozen, 4), "ba_iter_ms": round(ba_ms, 4),
            "ms_per_frame_from_iteration_times": round(total, 3),
            "formula": "iter_RO*ro + tracking.iter*go + mapping.iters*ba/map_every (FastCaMo-synth cadence 5/10/15/3)"}


def measured_sequence(n_frames, dev, stream, samplers=("reference", "device"), **seq_kw):
    """tracking + mapping ms/frame MEASURED over a synthetic sequence at the reference cadence (5 RO rounds, 10
    tracking iterations, 15 mapping iterations every 3rd frame, keyframe every 15th, 500 initialisation iterations):
    mipsfusion_amd/sequence.py.  `reference`: pixel / keyframe-ray indices and jitter from the reference's own host
    generators (bit-identical index stream, drawn map_every frames ahead by producer threads); `device`: the same draws made
    on the GPU."""
    from mipsfusion_amd import sequence
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from run_sequence import trajectory
    out = {}
    for sampler in samplers:
        random.seed(0), np.random.seed(0), torch.manual_seed(0)
        cfg = synth.config_reference_defaults()                   # S = 50 + 25, sample 1800 + pixels_cur 800: as shipped
        gt = trajectory(cfg, n_frames)
        frames = [synth.make_frame(cfg, gt[k], seed=k, frame_id=k) for k in range(n_frames)]
        seq = sequence.GraphedSequence(cfg, dev, frames, kf_every=15, sampler=sampler, stream=stream, **seq_kw)
        res = seq.run(gt)
        out[sampler] = sequence.summarise(res, gt, cfg, "hipGraph replay per tracking frame / per BA round; RandomOptimizer rounds of a frame in one replay")
        out[sampler]["decoder_arithmetic"] = {"model": seq.model.decoder_precision, "random_optimizer": seq.ro.decoder_precision}
        del seq
        torch.cuda.empty_cache()
    return out


def measured_config3(n_frames, dev, stream):
    """BASELINE config 3: the full-size online loop over SEVERAL sub-maps -- a 300-frame walk through two rooms
    (synth.two_room_sequence), a new sub-map at the first keyframe behind the door (parameter store, recover_initial_param,
    fresh map optimiser, 500 initialisation iterations), a switch back to sub-map 0 on the return (store, load, 15 pose-only
    iterations of local_BA_switch).  Every frame counts, switch frames included."""
    from mipsfusion_amd import sequence
    random.seed(0), np.random.seed(0), torch.manual_seed(0)
    cfg = synth.config_two_rooms()
    t0 = time.perf_counter()
    gt, frames, schedule = synth.two_room_sequence(cfg, n_frames, kf_every=cfg["mapping"]["keyframe_every"])
    log(f"config 3: {n_frames} two-room frames rendered on the host in {time.perf_counter() - t0:.1f}s, schedule {schedule}")
    seq = sequence.GraphedSequence(cfg, dev, frames, kf_every=cfg["mapping"]["keyframe_every"], sampler="reference", stream=stream,
                                   schedule=schedule)
    res = seq.run(gt)
    out = sequence.summarise(res, gt, cfg, "hipGraph replay per tracking frame / BA round / 25 initialisation iterations / switch refinement")
    out["decoder_arithmetic"] = {"model": seq.model.decoder_precision, "random_optimizer": seq.ro.decoder_precision}
    out["sampler"] = ("reference host generators (run ahead by producer threads) for tracking and local BA; the sub-map "
                      "initialisation and switch-refinement iterations draw pixels and jitter on the device")
    out.pop("frame_ms_all", None)
    del seq
    torch.cuda.empty_cache()
    return out


def multi_gpu_checks(cfg, model, dev, rank, world):
    """N > 1 only (every rank calls this): the two other shardings of SURVEY 8e, exercised over the real process group.
    (row 3) RandomOptimizer particle split: a replica of one sub-map on every rank, 2000 particles / world per rank, one
    all_gather of [2000, 8] per round -- must reproduce the unsplit pose exactly.  (row 1) cross-sub-map global BA
    (InactiveMap.py:375-474) over the ranks' OWN sub-maps: prediction-table all-reduce + the (n-1) x 7 pose-gradient
    all-reduce per pose step -- every rank must end with the same anchors."""
    import types
    import torch.distributed as dist
    from mipsfusion_amd.RandomOptimizer import RandomOptimizer
    from mipsfusion_amd.global_ba import PairTerm, ShardedGlobalBA, frozen, model_query
    out = {}
    # ---- row 3: particle split
    rcfg = cfg["tracking"]["RO"]
    rcfg.setdefault("initial_scaling_factor", 0.02)
    rcfg.setdefault("rescaling_factor", 0.5)
    cfg["tracking"].setdefault("ignore_edge_W", 20), cfg["tracking"].setdefault("ignore_edge_H", 20)
    H, W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
    frame = synth.make_frame(cfg, seed=1)
    ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frame["direction"])
    replica, _, _ = build_submap(cfg, dev, seed=0)              # the same sub-map on every rank
    replica.eval()
    np.random.seed(0)
    ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
    init = frame["c2w"].clone()
    init[:3, 3] += torch.tensor([0.02, -0.015, 0.01])
    n_ro = max(1, cfg["tracking"]["iter_RO"])

    def timed_ro(split):
        ro.particle_split = split
        for _ in range(2):
            pose = ro.optimize(replica, frame["depth"], init, None, n_iter=n_ro)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            pose = ro.optimize(replica, frame["depth"], init, None, n_iter=n_ro)
        torch.cuda.synchronize()
        return pose, (time.perf_counter() - t0) / 5 / n_ro * 1e3
    # ONE comparison, reported as it came out (round 3 retried it when ranks shared a device: the differences it saw there are
    # the packed-fp32 hazard of DESIGN.md 4h, fixed in the kernels; the line says which topology this was)
    pose_one, ms_one = timed_ro(False)
    pose_split, ms_split = timed_ro(True)
    out["ranks_share_a_device"] = bool(world > 1 and torch.cuda.device_count() < world)
    out["ro_round_ms_unsplit"] = round(mdist.max_over_ranks(ms_one, dev), 4)
    out["ro_round_ms_particle_split"] = round(mdist.max_over_ranks(ms_split, dev), 4)
    out["ro_split_pose_equals_unsplit"] = bool(torch.equal(pose_one, pose_split))
    out["ro_split_pose_max_abs_diff"] = float((pose_one.double() - pose_split.double()).abs().max())
    # diagnostics of the check itself: the replicas and the unsplit poses must agree over the ranks
    sig = torch.stack([p.detach().double().sum() for p in replica.parameters()] + [pose_one.to(dev).double().sum()]).to(dev)
    sigs = mdist.all_gather_ragged(sig[None].float().repeat(1, 1), world, world) if world > 1 else sig[None].float()
    out["ro_replica_and_unsplit_pose_spread_over_ranks"] = float((sigs - sigs[:1]).abs().max())
    out["ro_particles_per_rank"] = ro.particle_size // world
    del replica
    # ---- row 1: global BA over the ranks' own sub-maps (sub-map id = rank), chain of adjacent pairs
    anchors = torch.eye(4, device=dev)[None].repeat(world, 1, 1)
    for s in range(1, world):
        anchors[s, :3, 3] = torch.tensor([0.03 * s, -0.02 * s, 0.01 * s], device=dev)
    g = torch.Generator().manual_seed(1234)                     # the same term batches on every rank
    bs = max(cfg["mapping"]["sample"] // max(1, world - 1), cfg["mapping"]["sample"] // 4)
    f7 = torch.cat([frame["direction"], frame["rgb"], frame["depth"][..., None]], -1).reshape(-1, 7)
    n_iter = 20                                                 # InactiveMap.py:408
    batches = []
    for _ in range(n_iter):
        terms = []
        for i in range(world - 1):
            idx = torch.randint(0, f7.shape[0], (bs,), generator=g)
            terms.append(PairTerm(i, i + 1, f7[idx].to(dev), frame["c2w"][None].to(dev), 5.0))
        batches.append(terms)
    model.eval()
    with frozen([model]):
        ba = ShardedGlobalBA(model_query({rank: model}), [rank], anchors, cfg["training"]["trunc"],
                             cfg["mapping"]["lr_rot"], cfg["mapping"]["lr_trans"], cfg["mapping"]["pose_accum_step"])
        ba.iteration(batches[0])
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for terms in batches[1:]:
            loss = ba.iteration(terms)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (n_iter - 1) * 1e3
    model.train()
    res = ba.result()
    spread = mdist.exchange_poses(res[:, :3, :3].reshape(world, 9)[:, :4], res[:, :3, 3])   # any 7 numbers per anchor
    # ---- row 2: ray-data-parallel TRAINING of one replicated sub-map (mipsfusion_amd/ray_dp.py): every rank renders its
    #      share of the SAME 4096-ray batch; grid gradient reduce-scatter -> Adam on this rank's 1/world slice -> all-gather
    from mipsfusion_amd.ray_dp import RayDataParallelStep
    rep, rframes, rposes = build_submap(cfg, dev, seed=0)
    rep.train()
    mpc = cfg["mapping"]
    rdp = RayDataParallelStep(
        rep, lambda shard: FusedAdam([{"params": [shard], "eps": 1e-15, "lr": mpc["lr_embed"]}], betas=(0.9, 0.99)),
        lambda ps: FusedAdam([{"params": ps, "weight_decay": 1e-6, "lr": mpc["lr_decoder"]}], betas=(0.9, 0.99)))
    gq = torch.Generator().manual_seed(77)                      # the same batches on every rank
    fr = rframes[-1]
    f7r = torch.cat([fr["direction"], fr["rgb"], fr["depth"][..., None]], -1).reshape(-1, 7)
    c2w = rposes[-1].to(dev)
    n_rdp, losses_rdp = 12, []
    b_, e_ = rdp.my_share(N_RAYS)
    t_rdp = None
    for it in range(n_rdp):
        idx = torch.randint(0, f7r.shape[0], (N_RAYS,), generator=gq)
        rays = f7r[idx][b_:e_].to(dev)
        noise = torch.rand(N_RAYS, N_SAMPLES, generator=gq)[b_:e_].to(dev)
        rays_d = torch.sum(rays[:, None, :3] * c2w[:3, :3], -1)
        rays_o = c2w[None, :3, 3].repeat(e_ - b_, 1)
        if it == 2:
            torch.cuda.synchronize()
            dist.barrier()
            t_rdp = time.perf_counter()
        ret = rep.forward(rays_o, rays_d, rays[:, 3:6].contiguous(), rays[:, 6:7].contiguous(), noise=noise)
        loss_rdp = get_loss_from_ret(ret, cfg["training"])
        loss_rdp.backward()
        rdp.step()
        losses_rdp.append(float(loss_rdp))
    torch.cuda.synchronize()
    ms_rdp = (time.perf_counter() - t_rdp) / (n_rdp - 2) * 1e3
    chk = torch.stack([rep.embed_fn.params.detach().double().sum(), rep.embed_fn.params.detach().double().abs().sum(),
                       torch.cat([p.detach().reshape(-1) for p in rep.decoder.parameters()]).double().sum()])
    sums = [torch.empty_like(chk) for _ in range(world)] if dist.get_backend() != "gloo" else None
    if sums is None:
        h = chk.cpu()
        sums = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(sums, h)
    else:
        dist.all_gather(sums, chk)
    out["ray_dp_training"] = {"ms_per_step": round(mdist.max_over_ranks(ms_rdp, dev), 4), "rays_per_rank": e_ - b_,
                              "params_equal_over_ranks": bool(all(torch.equal(s.cpu(), sums[0].cpu()) for s in sums)),
                              "loss_first_last": [round(losses_rdp[0], 5), round(losses_rdp[-1], 5)],
                              "collectives_per_step": f"reduce_scatter[{rep.embed_fn.params.numel()} floats] + all_gather[same] "
                                                      f"+ all_reduce[{sum(p.numel() for p in rep.decoder.parameters())} floats]"}
    del rep, rdp
    torch.cuda.empty_cache()
    # the two collectives of the sharded paths on their own (latency-bound: K x 7 and (n - 1) x 7 floats), 50 calls each
    def timed_collective(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return mdist.max_over_ranks((time.perf_counter() - t0) / n * 1e3, dev)
    k_rot, k_trans = torch.randn(4, 4, device=dev), torch.randn(4, 3, device=dev)
    out["pose_all_gather_ms"] = round(timed_collective(lambda: mdist.exchange_poses(k_rot, k_trans)), 4)
    g7 = torch.zeros(max(1, world - 1), 7, device=dev)
    out["pose_grad_all_reduce_ms"] = round(timed_collective(lambda: mdist.all_reduce_sum_(g7)), 4)
    tbl = torch.zeros(max(1, world - 1), 2, bs, device=dev)
    out["prediction_table_all_reduce_ms"] = round(timed_collective(lambda: mdist.all_reduce_sum_(tbl)), 4)
    out["backend"] = dist.get_backend()
    out["global_ba_iter_ms"] = round(mdist.max_over_ranks(ms, dev), 4)
    out["global_ba_final_loss"] = float(loss)
    out["global_ba_anchor_spread_over_ranks"] = float((spread - spread[0:1]).abs().max())
    out["global_ba_anchors_moved"] = bool((res[1:] - anchors[1:]).abs().max() > 1e-5)
    out["global_ba_collectives_per_pose_step"] = (f"{cfg['mapping']['pose_accum_step']} x all_reduce[{world - 1} pairs, 2, {bs}] "
                                                  f"+ 1 x all_reduce[{world - 1} x 7]")
    return out


def cpu_baseline(cfg, loop, n_rays, iters, warmup=3):
    """The oracle (torch-CPU restatement of the reference path, oracle/path_cpu.py) on a bounded sample."""
    from oracle import path_cpu
    torch.set_num_threads(usable_cores())
    cpu = path_cpu.CpuScene(cfg, cfg["mapping"]["bound"], cfg["mapping"]["localMLP_max_len"])
    cpu.load_state_dict({k: v.cpu() for k, v in loop.model.state_dict().items()})
    opt = torch.optim.Adam([{"params": cpu.decoder.parameters(), "weight_decay": 1e-6, "lr": 0.01},
                            {"params": cpu.embed_fn.parameters(), "eps": 1e-15, "lr": 0.01}], betas=(0.9, 0.99))
    rays, owner = loop.batch(0)
    rays, owner = rays[:n_rays].cpu(), owner[:n_rays].cpu()
    poses = torch.cat([loop.pose_fixed, qt_to_transform_matrix(loop.cur_rot, loop.cur_trans)], 0).detach().cpu()
    noise = loop.noise[0][:n_rays].cpu()
    rays_d = torch.sum(rays[:, :3][..., None, :] * poses[owner, :3, :3], -1)
    rays_o = poses[owner, :3, -1]

    def it():
        opt.zero_grad()
        ret = cpu.train_forward(rays_o, rays_d, rays[:, 3:6], rays[:, 6:7], noise, 0.01)
        loss = path_cpu.total_loss(ret, cfg["training"])
        loss.backward()
        grid_grad = cpu.embed_fn.params.grad.clone()
        opt.step()
        return ret, loss.detach(), grid_grad
    # the same iteration on the GPU from the same parameters: the checker's verdict next to its speed
    dev = loop.dev
    m = loop.model
    m.zero_grad(set_to_none=True)
    g_ret = m.forward(rays_o.to(dev), rays_d.to(dev), rays[:, 3:6].contiguous().to(dev), rays[:, 6:7].contiguous().to(dev),
                      EMD_w=0.01, noise=noise.to(dev))
    g_loss = path_cpu.total_loss(g_ret, cfg["training"])
    g_loss.backward()
    t_w = time.perf_counter()
    c_ret, c_loss, c_grad = it()            # (the first warm-up iteration is the one compared with the GPU below)
    log(f"cpu_baseline warm-up iteration {time.perf_counter() - t_w:.1f}s on {torch.get_num_threads()} threads")

    def rel(a, b):
        return float((a.detach().cpu().double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    parity = {"loss_rel_err": abs(float(g_loss) - float(c_loss)) / abs(float(c_loss)),
              "depth_map_rel_err": rel(g_ret["depth"], c_ret["depth"]), "rgb_map_rel_err": rel(g_ret["rgb"], c_ret["rgb"]),
              "grid_grad_rel_err": rel(m.embed_fn.params.grad, c_grad)}
    log("cpu_baseline: GPU vs oracle on this batch " + ", ".join(f"{k} {v:.2e}" for k, v in parity.items()))
    m.zero_grad(set_to_none=False)
    for _ in range(max(0, warmup - 1)):         # 3 warm-up + 10 timed iterations
        it()
    t0 = time.perf_counter()
    done = 0
    for _ in range(iters):
        it()
        done += 1
        if time.perf_counter() - t0 > 45.0:          # keep the default run within minutes on slow hosts
            break
    iters = done
    dt = (time.perf_counter() - t0) / iters
    return {"value": n_rays * N_SAMPLES / dt, "unit": "rays*samples/s", "cores": torch.get_num_threads(),
            "kind": "port", "s_per_iter": round(dt, 3),
            "gpu_vs_oracle_same_batch": {k: float(f"{v:.3e}") for k, v in parity.items()},
            "sample": f"{warmup} warm-up + {iters} timed full iterations (fwd+bwd+dense Adam over the 2^19 grid) of oracle/path_cpu.py on "
                      f"{n_rays} of the 4096 rays x 64 samples of the same batch, torch CPU threads = cores"}


def pmc_traffic(key="traffic_bytes_per_launch"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_latest.json, written by
    tools/pmc_to_json.py from `tools/pmc.sh`; FETCH_SIZE/WRITE_SIZE are in KiB).  `traffic` applies the guide's gfx950
    correction -- FETCH_SIZE reports half the bytes of a 16-byte-per-lane streaming read -- to the kernels that read
    that way (decoder kernels, Adam, hashgrid_dx; calibrated on adam_kernel: 72 MB reported for 144 MB read);
    `traffic_uncorrected` is the plain FETCH_SIZE + WRITE_SIZE sum."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f).get(key, {})


def kernel_table(prof, M, precision, live_share, pair_share, with_traffic=True):
    """per-kernel roofline rows from the event pairs of an eager pass (`prof`: name -> (launches, mean ms)) and the dominant
    kernel's row as `roofline`.  `achieved` = algorithmic bytes (or FLOP) of the units a launch PROCESSED / its mean duration."""
    kernels = {}
    traffic = pmc_traffic() if with_traffic else {}
    traffic_raw = pmc_traffic("traffic_bytes_per_launch_uncorrected") if with_traffic else {}
    cost = kernel_cost(precision)
    hashgrid_route_ms = None
    if "hashgrid_route" in prof and "hashgrid_bwd" in prof:
        # the routing half of the scatter runs on a second stream next to the forward pass: its kernels' time is ADDED to
        # the backward's for the roofline (the work is done, wherever it runs); `overlapped_ms` says how much is hidden
        n_b, ms_b = prof["hashgrid_bwd"]
        route_ms = prof["hashgrid_route"][1]
        prof = dict(prof)
        prof["hashgrid_bwd"] = (n_b, ms_b + route_ms)
        del prof["hashgrid_route"]
        hashgrid_route_ms = route_ms
    shares = {"tile": live_share, "pair": pair_share, None: 1.0}
    for name, (n_launch, ms) in prof.items():
        mult, share, design = None, 1.0, 0.0
        hbm_view = None
        if name in cost:
            bound, per_live, per_dead, liveness, dtype, mult, design = cost[name][:7]
            share = shares[liveness] if shares[liveness] is not None else 1.0
            work = M * (share * per_live + (1.0 - share) * per_dead)
            work_full = M * per_live
            # `achieved` counts ALGORITHMIC FLOPs (one fp32-exact product = 2 FLOP, however many 16-bit MFMA
            # products carry it); the issued figure (x mult) sits next to it as `frac_issued`
            if len(cost[name]) > 7:
                hbm_bytes = M * (share * cost[name][7] + (1.0 - share) * cost[name][8])
                hbm_view = {"algorithmic_bytes_per_launch": round(hbm_bytes, 1), "achieved_GBps": round(hbm_bytes / (ms * 1e-3) / 1e9, 2),
                            "frac_of_8_TBps": round(hbm_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        elif name == "adam_step":
            bound, dtype, liveness = "hbm", "f32", None
            work = work_full = 28.0 * N_GRID_PARAMS
        else:
            continue
        if bound == "hbm":
            scale, peak, unit = 1e9, HBM_PEAK_GBS, "GB/s"
        else:
            scale, peak, unit = 1e12, (MFMA_F32_PEAK_TFLOPS if (mult or 1) == 1 else MFMA_F16_PEAK_TFLOPS), "TFLOP/s"
        achieved = work / (ms * 1e-3) / scale
        kernels[name] = {"bound": bound, "achieved": round(achieved, 2), "peak": peak, "unit": unit,
                         "frac": round(achieved / peak, 4), "avg_ms": round(ms, 4), "dtype": dtype,
                         "launches": n_launch, "work_per_launch": round(work, 1),
                         "traffic": traffic.get(name), "traffic_uncorrected": traffic_raw.get(name)}
        if bound == "mfma" and mult and mult > 1:
            kernels[name]["frac_issued"] = round(achieved * mult / peak, 4)
            kernels[name]["frac_of_fp32_mfma_peak"] = round(achieved / MFMA_F32_PEAK_TFLOPS, 4)
            kernels[name]["products_issued_per_algorithmic_product"] = mult
            kernels[name]["achieved_is"] = (f"algorithmic FLOP/s of the units processed against the dense 16-bit peak; every fp32-exact "
                                            f"product is issued as {mult} 16-bit MFMA products (frac_issued)")
        if hbm_view is not None:
            kernels[name]["hbm_view"] = hbm_view
        if design:
            kernels[name]["design_bytes_per_launch"] = round(M * design, 1)
            kernels[name]["design_note"] = "bytes the design moves on top of the algorithmic ones (the Jacobian kept for the backward), not priced"
        if liveness is not None:
            kernels[name]["units"] = f"live {liveness}s: share {share:.4f} of the batch"
            kernels[name]["frac_full_batch"] = round(work_full / (ms * 1e-3) / scale / peak, 4)
        if mult is not None:        # matrix-pipe view of the decoder kernels (products of the units processed)
            issued = DECODER_FLOP_PER_SAMPLE * M * share * mult / (ms * 1e-3) / 1e12
            mpeak = MFMA_F32_PEAK_TFLOPS if mult == 1 else MFMA_F16_PEAK_TFLOPS
            kernels[name]["matrix_pipe"] = {"algorithmic_tflops": round(issued / mult, 2), "issued_tflops": round(issued, 2),
                                            "products_issued_per_algorithmic_product": mult,
                                            "peak_tflops": mpeak, "utilisation": round(issued / mpeak, 4)}
        if name == "hashgrid_bwd" and hashgrid_route_ms is not None:
            kernels[name]["routing_on_second_stream_ms"] = round(hashgrid_route_ms, 4)
    dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"]) if kernels else None
    roofline = dict(kernels[dominant], kernel=dominant) if dominant else None
    if roofline is not None:
        roofline["traffic_source"] = ("profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command on an earlier box (per launch, gfx950 x2 FETCH correction where the kernel "
                                      "reads 16 B per lane); not collected in this run")
        roofline["device_stream_ceilings_TBps"] = STREAM_CEILINGS_TBS
    return kernels, roofline


LINE_LIMIT = 6000            # the driver keeps an 8 000-character stdout tail: the line it parses has to fit with room to spare
REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out):
    """The ONE line the driver parses, built from the full result `out`: the contract fields, `roofline` (dominant kernel,
    algorithmic AND issued fractions), `cpu_baseline`, the frame metric, the unchanged caller's step and one row of
    (avg_ms, frac) per kernel.  Everything else -- variants, per-kernel detail, sequences, inference -- goes to
    bench_detail.json beside this script and to stderr.  Always shorter than LINE_LIMIT (tests/test_host_cpu.py)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"))
    line["vs_baseline"] = out.get("vs_baseline")                  # null is the contract's value when nothing is published
    cfg = out.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "rays", "samples_per_ray", "hash_size", "parallelism"))
    line["launch"] = out.get("launch")
    line["eager_ms_per_step"] = out.get("eager_ms_per_step")
    if out.get("step_ms_stats"):
        line["step_ms_stats"] = _pick(out["step_ms_stats"], ("steps", "min", "median", "p95", "max"))
    if out.get("value_at_median_step") is not None:
        line["value_at_median_step"] = out["value_at_median_step"]
    line["board"] = _pick(out.get("board") or {}, ("sclk_mhz_median", "sclk_mhz_min", "power_w_median", "samples")) or None
    roof = out.get("roofline")
    if roof:
        r = _pick(roof, ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_ms", "launches", "work_per_launch", "traffic",
                         "frac_issued", "frac_of_fp32_mfma_peak", "products_issued_per_algorithmic_product", "units",
                         "frac_full_batch"))
        r["traffic"] = roof.get("traffic")                         # null when no PMC pass is on file
        if roof.get("hbm_view"):
            r["hbm_view_frac_of_8_TBps"] = roof["hbm_view"].get("frac_of_8_TBps")
        r["traffic_source"] = "profiles/pmc_latest.json (rocprofv3 --pmc passes of this command, per launch; not collected in this run)"
        line["roofline"] = r
    else:
        line["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "s_per_iter", "gpu_vs_oracle_same_batch"))
        c["value"] = round(float(cb["value"]), 1)
        c["sample"] = str(cb.get("sample", ""))[:200]
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    kern = out.get("kernels") or {}
    line["kernels_avg_ms_frac"] = {k: [v.get("avg_ms"), v.get("frac")] for k, v in kern.items()}
    if out.get("forward_only"):
        line["forward_only"] = _pick(out["forward_only"], ("value", "ms"))
    frame = out.get("frame") or {}
    if frame:
        f = _pick(frame, ("tracking_plus_mapping_ms_per_frame", "ms_per_frame_device_sampling", "ms_per_frame_from_iteration_times",
                          "ro_iter_ms", "go_iter_ms", "ba_iter_ms", "decoder_arithmetic"))
        c3 = frame.get("config3_multi_submap")
        if c3:
            f["config3_multi_submap"] = _pick(c3, ("frames", "ms_per_frame_mean", "ate_rmse_m"))
        line["frame"] = f
    var = out.get("variants") or {}
    if var:
        line["variants_ms_per_step"] = {k: v.get("ms_per_step") for k, v in var.items() if isinstance(v, dict)}
        uc = var.get("unchanged_caller") or {}
        if uc.get("ms_per_step"):
            line["unchanged_caller_ms_per_step"] = uc["ms_per_step"]
            line["unchanged_caller_over_headline"] = round(uc["ms_per_step"] / out["ms_per_step"], 2)
    if out.get("gradient_sparsity"):
        line["gradient_sparsity"] = _pick(out["gradient_sparsity"], ("live_32_sample_tiles", "live_sample_level_pairs"))
    mg = out.get("multi_gpu")
    if mg:
        line["multi_gpu"] = _pick(mg, ("ms_per_step_of_each_rank", "pose_all_gather_ms", "pose_grad_all_reduce_ms",
                                       "prediction_table_all_reduce_ms", "backend", "global_ba_iter_ms"))
        if mg.get("ranks"):
            line["multi_gpu"]["ranks"] = _pick(mg["ranks"], ("world_size", "backend", "distinct_devices"))
    line["scaling_curve"] = str(out.get("scaling_curve", ""))[:160]
    line["detail"] = "bench_detail.json (full result: variants, per-kernel rows, sequences, inference); also on stderr"
    text = json.dumps(line, separators=(",", ":"))
    # never over the limit: drop the optional blocks, least important first
    for k in ("gradient_sparsity", "step_ms_stats", "variants_ms_per_step", "kernels_avg_ms_frac", "multi_gpu", "forward_only", "frame"):
        if len(text) < LINE_LIMIT:
            break
        line.pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def emit(out):
    """full result -> bench_detail.json + stderr; the compact line -> stdout (last line).  Rank 0 only: the other ranks' results
    go to stderr (the launcher forwards ONE line, and the detail file must be that rank's)."""
    if int(os.environ.get("RANK", "0")) != 0:
        sys.stderr.write(f"[bench detail rank {os.environ.get('RANK')}] " + json.dumps(out) + "\n")
        sys.stderr.flush()
        return
    detail = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError as e:
                log(f"bench_detail.json not written in {d}: {e}")
    sys.stderr.write("[bench detail] " + detail + "\n")
    sys.stderr.flush()
    print(compact_line(out), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) before this process has touched the GPU, pass rank 0's JSON line through, and leave with
    the children's exit code.  (Never an exec: a process that has initialised the GPU must not replace itself.)"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL / cross-process GPU memory need it on this pool
    for k in ("OMP_NUM_THREADS",):                              # every rank sizes its own pools (bench.py top)
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:9])} ...")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last_json = None
    for line in proc.stdout:
        if line.startswith("{"):
            last_json = line.rstrip("\n")
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(f"a rank failed (torch.distributed.run exit code {rc})")
    if last_json is None:
        raise SystemExit("the ranks printed no JSON line")
    print(last_json, flush=True)
    raise SystemExit(0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                                  # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    backend = os.environ.get("MIPSF_BENCH_BACKEND", "nccl")      # "gloo" only to debug N>1 on a single GPU
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} over RCCL needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(MIPSF_BENCH_BACKEND=gloo shares a device between ranks: a debugging mode, not a measurement)")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    ranks_info = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        assert dist.get_backend() == backend
        # which device every rank really sits on: (bus id, uuid) gathered over the group
        props = torch.cuda.get_device_properties(local)
        me = f"{local}:{getattr(props, 'pci_bus_id', '?')}:{getattr(props, 'uuid', '?')}"
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        ranks_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend() + (" (= RCCL on ROCm)" if backend == "nccl" else ""),
                      "device_of_rank": everyone, "distinct_devices": len(set(everyone))}
        if backend == "nccl":
            assert ranks_info["distinct_devices"] == world, f"ranks share devices: {everyone}"

    stream = work_stream(dev)        # everything (setup, eager pass, capture, replays) runs on this one stream
    cfg = synth.config_headline()
    log(f"rank {rank}/{world} on {torch.cuda.get_device_name(local)}; building submap")
    model, frames, poses = build_submap(cfg, dev, seed=rank)
    table, db, R = build_ray_table(cfg, frames, dev)
    use_graph = not args.no_graph and not args.torch_pose and args.steps % cfg["mapping"]["pose_accum_step"] == 0
    # one fresh batch for every step of every pass (setup, warm-ups, eager pass, graph pass, stats pass); beyond 1024
    # sets the sequence wraps around (a 4 ms host draw per set would otherwise dominate the run time of long benches)
    n_sets = min(1024, args.setup_iters + 3 * args.warmup + 2 * args.steps + args.stats_steps + 16)
    t_draw = time.perf_counter()
    idx_rows, idx_owner = draw_index_sets(cfg, frames, db, R, n_sets)
    log(f"{n_sets} index sets drawn on the host with the reference samplers in {time.perf_counter() - t_draw:.1f}s")
    loop = MappingLoop(cfg, model, poses, table, idx_rows, idx_owner, dev, torch_pose=args.torch_pose,
                       capturable=use_graph)
    log("setup iterations")
    for _ in range(args.setup_iters):
        loop.step()
    torch.cuda.synchronize()
    log("setup done")

    def exchange_poses():
        # after each BA round every submap publishes its optimised keyframe poses (RCCL all_gather)
        mdist.exchange_poses(loop.cur_rot, loop.cur_trans)

    for _ in range(args.warmup):
        loop.step()
    exchange_poses()
    ba_round = cfg["mapping"]["iters"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- eager pass: K steps with per-kernel event pairs on the launch stream (kernel durations for the roofline)
    ops.PROFILE = {}
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        loop.step()
        if (k + 1) % ba_round == 0:
            exchange_poses()
    barrier()
    eager_elapsed = time.perf_counter() - t0
    prof = ops.profile_summary()
    ops.PROFILE = None
    log(f"eager timed region: {args.steps} steps in {eager_elapsed * 1e3:.1f} ms")
    elapsed, step_stats, board = eager_elapsed, None, None

    # ---- graph pass (the number reported as `value`): the same K steps as hipGraph replays of pose_accum_step
    #      iterations; fresh ray batches are copied into the static input buffers before every replay
    if use_graph:
        n_inner = cfg["mapping"]["pose_accum_step"]
        loop.make_static(n_inner)
        loop.refill_static()
        graphed = GraphedSteps(loop.step_static, n_inner, stream=stream)
        for _ in range(max(1, args.warmup // n_inner)):
            loop.refill_static()
            graphed.replay()
        barrier()
        board_sampler = BoardSampler(dev.index or 0).__enter__()
        t0 = time.perf_counter()
        for r in range(args.steps // n_inner):
            loop.refill_static()
            graphed.replay()
            if ((r + 1) * n_inner) % ba_round == 0:
                exchange_poses()
        barrier()
        elapsed = time.perf_counter() - t0
        log(f"graph timed region: {args.steps} steps in {elapsed * 1e3:.1f} ms")
        # ---- spread: further steps, every replay (= n_inner steps incl. its input refill) bracketed by an event pair
        n_rep = max(1, args.stats_steps // n_inner)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_rep)]
        for a, b in ev:
            a.record()
            loop.refill_static()
            graphed.replay()
            b.record()
        torch.cuda.synchronize()
        board_sampler.__exit__()
        board = board_sampler.summary()
        per_step = np.array([a.elapsed_time(b) / n_inner for a, b in ev])
        step_stats = {"steps": n_rep * n_inner, "granularity": f"one replay of {n_inner} steps incl. its input refill",
                      "min": round(float(per_step.min()), 4), "median": round(float(np.median(per_step)), 4),
                      "p95": round(float(np.percentile(per_step, 95)), 4), "max": round(float(per_step.max()), 4)}
    # ---- the same step under the two switches the headline depends on, and the unchanged caller (N = 1 only)
    variants = None
    headline_precision = model.decoder_precision
    if use_graph and world == 1 and not args.no_variants:
        variants = {}
        M_ = N_RAYS * N_SAMPLES

        def variant(prec, what):
            """the same graph-replayed step at another decoder arithmetic, with its own eager pass for the kernel table"""
            model.decoder_precision = prec
            for _ in range(3):
                loop.step()
            ops.PROFILE = {}
            for _ in range(20):
                loop.step()
            torch.cuda.synchronize()
            vprof = ops.profile_summary()
            ops.PROFILE = None
            vk, vroof = kernel_table(vprof, M_, prec, ops.last_live_tile_share() if prec != "f32" else None,
                                     ops.last_live_record_share(), with_traffic=False)
            ms = graphed_ms_per_step(loop, stream, n_inner, args.steps, args.warmup)
            model.decoder_precision = headline_precision
            return {"ms_per_step": round(ms, 4), "value": round(M_ / (ms * 1e-3), 1), "dtype": what, "roofline": vroof,
                    "kernels": {k: v for k, v in vk.items() if k.startswith("decoder_")}}
        variants["decoder_precision_f16x3_fast_mode"] = variant(
            "f16x3", "fast mode (opt-in, JointEncoding.decoder_precision = 'f16x3'): f16 MFMA on hi/lo split operands, 22-23 "
                     "operand bits, lean records; 2-3x the fp32 kernels' error against fp64 truth")
        variants["decoder_precision_f32"] = variant(
            "f32", "f32 everywhere (fp32-input MFMA = exact fp32 products, the round-1 kernels: full records, no tile short cut)")
        ops.SKIP_ZERO_TILES = False                  # = MIPSF_NO_TILE_SKIP=1: chain, weight gradients and dx visit every tile
        ms = graphed_ms_per_step(loop, stream, n_inner, args.steps, args.warmup)
        variants["dense_no_tile_skip"] = {"ms_per_step": round(ms, 4), "value": round(M_ / (ms * 1e-3), 1),
                                          "note": "MIPSF_NO_TILE_SKIP=1: zero-gradient tiles are processed like live ones"}
        ops.SKIP_ZERO_TILES = True
        log(f"variants: { {k: v['ms_per_step'] for k, v in variants.items()} }")
        variants["unchanged_caller"] = unchanged_caller_rate(cfg, model, frames, poses, table, db, R, dev, min(args.steps, 20))
        log(f"unchanged caller: {variants['unchanged_caller']}")
    per_rank_ms = None
    if dist is not None:        # every rank's own step time (diagnosis of the first multi-GPU runs: which rank is slow)
        mine = torch.tensor([elapsed / args.steps * 1e3], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [round(float(t), 4) for t in allr]
    elapsed = mdist.max_over_ranks(elapsed, dev)
    multi = multi_gpu_checks(cfg, model, dev, rank, world) if world > 1 else None
    if multi is not None:
        multi["ms_per_step_of_each_rank"] = per_rank_ms

    if rank != 0:
        dist.barrier()              # rank 0 finishes its untimed extras, then everybody leaves together
        dist.destroy_process_group()
        return

    M = N_RAYS * N_SAMPLES
    ms_step = elapsed / args.steps * 1e3
    value = M * args.steps * world / elapsed

    live_share = ops.last_live_tile_share()
    pair_share = ops.last_live_record_share()
    kernels, roofline = kernel_table(prof, M, headline_precision, live_share, pair_share)

    fwd_rate, fwd_ms = forward_only_rate(model, loop, dev)
    log(f"forward-only {fwd_ms:.3f} ms")
    out = {
        "metric": "rays*samples/s per GPU, full optimisation iteration (fwd + bwd + pose+map Adam), 4096 rays x 64 samples, 640x480 RGB-D",
        "value": round(value, 1), "unit": "rays*samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
        "launch": "hipGraph replay of pose_accum_step iterations" if use_graph else "eager",
        "step_ms_stats": step_stats,
        "board": board,
        "value_at_median_step": None if not step_stats else round(M * world / (step_stats["median"] * 1e-3), 1),
        "batches": "a fresh ray batch every step: rows gathered in-step from the HBM ray table (4 keyframes x 30 000 rays "
                   "+ current frame) by host-drawn index sets (reference samplers), indices and jitter resident in HBM",
        "eager_ms_per_step": round(eager_elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE_OF[headline_precision],
        "data": "synthetic",
        "config": {"workload": "BASELINE config 2: FastCaMo-synth apartment_2, 1 active submap per GPU, 4096 rays x 64 samples, "
                               "hash grid 2^19 x 16 x 2, 620x460 synthetic RGB-D",
                   "rays": N_RAYS, "samples_per_ray": N_SAMPLES, "hash_size": cfg["grid"]["hash_size"],
                   "parallelism": f"submap-per-gpu x{world}" if world > 1 else "single gpu",
                   "ray_build": "torch eager ops" if args.torch_pose else "row gather + pose rays + sample placement in one kernel",
                   "host_cpus": f"{len(HOST_CPUS)} least-busy CPUs of one NUMA node ({HOST_CPUS[0]}..{HOST_CPUS[-1]})" if HOST_CPUS else "unconfined"},
        "forward_only": {"value": round(fwd_rate, 1), "unit": "rays*samples/s", "ms": round(fwd_ms, 4)},
        "decoder_gemm_forward": decoder_gemm_rates(model, dev, M),
        "variants": variants,
        "roofline": roofline, "kernels": kernels,
        "gradient_sparsity": {"live_32_sample_tiles": None if live_share is None else round(live_share, 4),
                              "live_sample_level_pairs": None if pair_share is None else round(pair_share, 4),
                              "note": "samples behind the truncation band carry no loss term and no rendering weight: "
                                      "exactly zero gradient; the backward chain, the weight-gradient kernel, dx and the grid "
                                      "scatter skip them (tiles / records); `kernels[*].achieved` counts the units a launch "
                                      "processed (live units at the full cost, dead units at the cost of finding them dead), "
                                      "`frac_full_batch` prices every unit as live"},
    }
    if not args.no_frame_estimate:
        out["frame"] = frame_estimate(cfg, model, loop, dev, ms_step, stream if use_graph else None)
        log("iteration-level frame numbers done")
        if args.seq_frames > 1 and use_graph:
            seq = measured_sequence(args.seq_frames, dev, stream)
            out["frame"]["measured_sequence"] = seq
            # the opt-in fast arithmetics next to it: f16x3 training, plain-f16 RandomOptimizer rounds (round 3's headline mode)
            fast = measured_sequence(args.seq_frames, dev, stream, samplers=("reference",), decoder_precision="f16x3", ro_precision="f16")
            out["frame"]["measured_sequence_fast_mode"] = fast["reference"]
            out["frame"]["ms_per_frame_reference_sampling"] = seq["reference"]["ms_per_frame_mean"]
            out["frame"]["ms_per_frame_device_sampling"] = seq["device"]["ms_per_frame_mean"]
            out["frame"]["tracking_plus_mapping_ms_per_frame"] = seq["reference"]["ms_per_frame_mean"]
            log("measured sequences done")
            if args.config3_frames > 1:
                out["frame"]["config3_multi_submap"] = measured_config3(args.config3_frames, dev, stream)
                log(f"config 3 done: {out['frame']['config3_multi_submap']['ms_per_frame_mean']} ms/frame")
        out["inference"] = inference_rates(cfg, model, dev)
        log("inference consumers done")
    if multi is not None:
        multi["ranks"] = ranks_info
        out["multi_gpu"] = multi
    out["scaling_curve"] = ("unmeasured: no multi-GPU node has been available to this build; N > 1 has run only as two gloo "
                            "ranks sharing one GPU (tests/test_gpu_configs.py)") if world == 1 else "this line is one point of it"
    if world == 1 and args.cpu_rays > 0:
        out["cpu_baseline"] = cpu_baseline(cfg, loop, args.cpu_rays, args.cpu_iters, args.cpu_warmup)
    emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
