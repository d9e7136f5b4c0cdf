"""The online loop around the hot path, with the host-side sampling OFF the critical path.

``MIPSFusion.run`` (mipsfusion.py:661-735) calls, per frame, ``tracking_render`` (:470-577) and every ``map_every``-th
frame ``local_BA`` (:259-371).  Both draw their pixel / keyframe-ray indices and the sample jitter on the HOST
(python ``random``, torch's default CPU generator: sampling_helper.py:20-68, keyframeSet.py:386-436, scene_rep.py:176),
once per optimisation iteration, between two GPU launches.  With the ~1 ms iterations of this package that host work
(4.7 ms per mapping iteration on 8 cores: ``randn_like`` over the 285 200 pixels, ``topk``, ``torch.rand(N, S)``,
``random.sample``) is 4x the GPU time.

Nothing in those draws depends on a GPU result: the call sequence and every argument (image size, ray counts, the
keyframes of the sub-map, the depth image of the frame) are known when the frame arrives.  So

* ``ReferenceSampleProducer`` runs the three generator streams in their own threads, one mapping period (map_every frames) AHEAD of the GPU, in the
  reference's exact call order (the torch stream: per tracking frame ``randn_like`` + ``iter`` x ``rand(n, S)``, per
  mapping iteration ``randn_like`` + ``rand(N, S)``; the python stream: per mapping iteration the ``random.sample`` calls
  of ``sample_rays_in_submap``; scoring + ``topk`` is generator-free and runs on a third thread) and leaves index sets
  and jitter in pinned buffers -- the index stream is bit-identical to calling the reference's functions in program
  order (tests/test_host_cpu.py::test_sample_producer_reproduces_the_sequential_index_stream);
* ``GraphedSequence`` replays one hipGraph per tracking frame and one per BA round whose static inputs (ray-table row
  indices, owning pose, jitter) are refilled from those buffers with three async copies.

``sampler="device"`` is the alternative without the reference's generators (indices and jitter drawn on the GPU: valid
depth only, without replacement, same per-keyframe shares).
"""
import os
import queue
import random
import sys
import threading
import time
import types
from typing import Dict, List, NamedTuple, Optional

import numpy as np
import torch

from . import hostrng, ops
from .helper_functions import sampling_helper as sh
from .helper_functions.geometry_helper import matrix_to_quaternion, qt_to_transform_matrix
from .helper_functions.utils import backward_from_one, get_loss_from_ret
from .keyframe_rays import DeviceRayDB


class FramePlan(NamedTuple):
    """Everything the host samplers of one frame depend on (all known when the frame arrives)."""
    frame_id: int
    depth: torch.Tensor                 # [H,W] CPU
    track: bool                         # tracking_render runs for this frame
    ba_kf_ids: Optional[torch.Tensor]   # related keyframe ids of the active sub-map if local_BA runs, else None
    kf_slot_base: int = 0               # row offset of the current frame's pixels in the device ray table


def packed(t, n):
    """[iters, n(, S)] view over the FRONT of the [iters, n_max(, S)] buffer `t`: the rays of a local-BA round vary with the
    number of related keyframes (8192, 6144, 5461 ...), and a [:, :n] slice of the full buffer is strided -- torch copies
    such a slice from pinned host memory through a pageable temporary (1.2-4 ms per round and a 10-20 ms first use,
    instead of three asynchronous DMA transfers)."""
    it = t.shape[0]
    inner = t.shape[2:]
    return t.view(-1)[:it * n * int(np.prod(inner, dtype=np.int64))].view(it, n, *inner)


class FrameSamples:
    """One ring slot of pinned buffers.  track_idx [n_track] (pixel index row*W+col), track_noise [iters, n_track, S];
    ba_packed() = ba_rows / ba_owner [iters, N] (ray-table row, index of the owning pose), ba_noise [iters, N, S] with
    N = n_ba, packed at the front of the slot's buffers."""

    def __init__(self, n_track, it_track, n_ba_max, it_ba, S, pinned):
        def buf(*shape, dtype=torch.float32):
            t = torch.empty(shape, dtype=dtype)
            return t.pin_memory() if pinned else t
        self.track_idx = buf(n_track, dtype=torch.int64)
        self.track_noise = buf(it_track, n_track, S)
        self.ba_rows = buf(it_ba, n_ba_max, dtype=torch.int64)
        self.ba_owner = buf(it_ba, n_ba_max, dtype=torch.int64)
        self.ba_noise = buf(it_ba, n_ba_max, S)
        self.n_ba, self.frame_id = 0, -1
        self.ready = threading.Event()
        self.free = threading.Event()
        self.free.set()
        self._pending, self._lock = 0, threading.Lock()

    def ba_packed(self):
        n = self.n_ba
        return packed(self.ba_rows, n), packed(self.ba_owner, n), packed(self.ba_noise, n)

    def _arm(self, parts):
        self._pending = parts
        self.ready.clear()

    def _part_done(self):
        with self._lock:
            self._pending -= 1
            if self._pending == 0:
                self.ready.set()


def ba_ray_counts(cfg, n_related):
    """(rays from stored keyframes, rays from the current frame) of one local-BA iteration (mipsfusion.py:295-309)."""
    mp, tk = cfg["mapping"], cfg["tracking"]
    if tk["iter_RO"] == 0:
        return mp["sample"], max(mp["sample"] // n_related, 50)
    return mp["sample"], max(mp["sample"] // n_related, mp["pixels_cur"])


_DIAG_OFF = os.environ.get("MIPSF_DIAG_STAGE_OFF", "")      # diagnosis: a stage that does no work (garbage samples)


class ReferenceSampleProducer:
    """See the module docstring.  ``submit(plan)`` enqueues a frame, ``get()`` returns the oldest submitted frame's
    ``FrameSamples`` once complete, ``release(s)`` hands the slot back after its contents were copied to the device."""

    def __init__(self, cfg, H, W, rays_per_kf, max_related, pinned=None, slots=3):
        self.cfg, self.H, self.W, self.R = cfg, H, W, rays_per_kf
        tr, tk, mp = cfg["training"], cfg["tracking"], cfg["mapping"]
        self.S = tr["n_samples_d"] + tr["n_range_d"]
        self.n_track, self.it_track, self.it_ba = tk["sample"], tk["iter"], mp["iters"]
        n_ba_max = max(sum(ba_ray_counts(cfg, k)) for k in range(1, max_related + 1))
        pinned = torch.cuda.is_available() if pinned is None else pinned
        self.slots = [FrameSamples(self.n_track, self.it_track, n_ba_max, self.it_ba, self.S, pinned) for _ in range(slots)]
        self._next = 0
        self._order: "queue.Queue[FrameSamples]" = queue.Queue()
        # per-pixel score draws go into a ring of preallocated buffers (see sampling_helper.draw_pixel_scores); the bounded
        # queue keeps the generator stage from lapping the top-k stage
        self._draw_bufs = [torch.empty(H * W, dtype=torch.float32) for _ in range(self.it_ba + 3)]
        self._draw_next = 0
        self._q_torch, self._q_topk, self._q_py = queue.Queue(), queue.Queue(maxsize=len(self._draw_bufs) - 2), queue.Queue()
        self._index_db = DeviceRayDB.__new__(DeviceRayDB)       # index arithmetic only (no storage, no gather)
        self._index_db.num_rays_to_save = rays_per_kf
        self._lattice_track = (cfg["sampling"]["n_rays_h"], cfg["sampling"]["n_rays_w"]) if "sampling" in cfg else \
            (tk["RO"]["n_rows"], tk["RO"]["n_cols"])
        self._threads = [threading.Thread(target=f, daemon=True, name=n) for f, n in
                         ((self._torch_stream, "mipsf-torch-rng"), (self._topk_stage, "mipsf-topk"),
                          (self._python_stream, "mipsf-python-rng"))]
        # The python-`random` stage is pure interpreter work: with CPython's default switch interval (5 ms) it keeps the GIL
        # for 5 ms at a time while the thread that drives the GPU waits for it between two launches -- measured: 5 ms of
        # `fill` per tracking frame and 15 ms per BA frame that are nothing but GIL waits.  0.1 ms hand-overs cost the
        # producers a few per cent and take those stalls away (restored by close()).
        self._switch_interval = sys.getswitchinterval()
        sys.setswitchinterval(min(self._switch_interval, float(os.environ.get("MIPSF_SWITCH_INTERVAL", "1e-4"))))
        # ... which does not help against the python-`random` stage: it hands the GIL over every ~0.15 ms anyway (it calls
        # into torch between its loops), and the consumer, whose 4x4 pose algebra is ~30 tiny torch calls each of which
        # drops and re-takes the GIL, queues behind it 30 times: `_set_pose` 0.15 -> 5 ms, a tracking frame 3.3 -> 12 ms on
        # the frames where the stage works (measured by switching the stages off one at a time, tools/micro/ab_stage.sh).
        # So the consumer can close this gate while it is busy on the host; the PYTHON stage (4 ms of work per frame,
        # the other two release the GIL inside their large torch ops) then pauses at its next chunk boundary and runs
        # while the consumer waits for the GPU (GraphedSequence.run opens the gate around its blocking read-backs).
        self.gate = threading.Event()
        self.gate.set()
        hostrng.available()                  # (its one-off self-check borrows the default generator: before the stages run)
        hostrng.topk_available(), hostrng.py_available()
        self._lattice_masks = {}
        for t in self._threads:
            t.start()
        self.host_ms = {"torch_rng": 0.0, "topk": 0.0, "python_rng": 0.0}

    # ------------------------------------------------------------------------------------------- client side
    def submit(self, plan: FramePlan):
        s = self.slots[self._next % len(self.slots)]
        self._next += 1
        s.free.wait()
        s.free.clear()
        s.frame_id = plan.frame_id
        iter_ro0 = self.cfg["tracking"]["iter_RO"] == 0
        n_topk = (1 if plan.track and not iter_ro0 else 0) + (self.it_ba if plan.ba_kf_ids is not None else 0)
        s._arm(1 + n_topk + 1)               # torch stream, every top-k item, python stream
        if plan.ba_kf_ids is not None:
            s.n_ba = sum(ba_ray_counts(self.cfg, plan.ba_kf_ids.shape[0]))
        else:
            s.n_ba = 0
        self._order.put(s)
        self._q_torch.put((plan, s))
        self._q_py.put((plan, s))

    def get(self, timeout=120.0) -> FrameSamples:
        s = self._order.get()
        if not s.ready.is_set():
            was_open = self.gate.is_set()
            self.gate.set()                  # the consumer has nothing better to do than wait: let the stages run
            ok = s.ready.wait(timeout)
            if not was_open:
                self.gate.clear()
            if not ok:
                raise RuntimeError("sample producer stalled")
        return s

    def release(self, s: FrameSamples):
        s.free.set()

    def close(self):
        for q in (self._q_torch, self._q_topk, self._q_py):
            q.put(None)
        sys.setswitchinterval(self._switch_interval)

    def _next_draw_buf(self, depth):
        if depth.numel() != self._draw_bufs[0].numel() or depth.dtype != torch.float32:
            return None                                       # unexpected image: a fresh tensor, as the reference does
        b = self._draw_bufs[self._draw_next % len(self._draw_bufs)]
        self._draw_next += 1
        return b

    # ------------------------------------------------------------------------------------------- the three stages
    def _torch_stream(self):
        """torch's default CPU generator, in the reference's call order."""
        # OpenMP's thread count is a per-thread setting whose default is the MACHINE's core count (256 on the GPU
        # hosts, whose cgroup grants 16): an unset count oversubscribes the quota and these draws run 30x slower.
        # The fills are serial under the generator lock anyway.
        torch.set_num_threads(1)
        iter_ro0 = self.cfg["tracking"]["iter_RO"] == 0
        while True:
            item = self._q_torch.get()
            if item is None:
                self._q_topk.put(None)
                return
            plan, s = item
            t0 = time.perf_counter()
            # the draws below go through the C replica of torch's CPU generator (mipsfusion_amd/hostrng.py: same stream,
            # same bits, checked against torch at start-up, torch's own functions otherwise): 0.55 instead of 1.5-2.3 ms
            # per mapping iteration of this serial stage
            if _DIAG_OFF == "torch":                                # diagnosis only (tools/micro): skip the draws
                n_items = (1 if plan.track and not iter_ro0 else 0) + (self.it_ba if plan.ba_kf_ids is not None else 0)
                for _ in range(n_items):
                    s._part_done()
                s._part_done()
                continue
            with hostrng.session() as g:
                if plan.track:
                    if not iter_ro0:                                # sample_pixels_mix (mipsfusion.py:519-523)
                        self._q_topk.put((plan, s, "track", 0, self._draw_scores(g, plan.depth)))
                    for i in range(self.it_track):                  # scene_rep.py:176, one draw per forward
                        g.rand_(s.track_noise[i])
                if plan.ba_kf_ids is not None:
                    noise = s.ba_packed()[2]
                    for i in range(self.it_ba):
                        self._q_topk.put((plan, s, "ba", i, self._draw_scores(g, plan.depth)))   # :302 / :306-307
                        g.rand_(noise[i])
            self.host_ms["torch_rng"] += (time.perf_counter() - t0) * 1e3
            s._part_done()

    def _draw_scores(self, g, depth):
        """sampling_helper.draw_pixel_scores (torch.randn_like of the depth image) into the next ring buffer"""
        buf = self._next_draw_buf(depth)
        if buf is None:                                       # unexpected image size / dtype: torch's own draw
            return g.through_torch(lambda: sh.draw_pixel_scores(depth))
        return g.randn_(buf)

    def _topk_stage(self):
        """Generator-free half of the valid-pixel samplers: mask, lattice blocking, top-k."""
        torch.set_num_threads(2)
        tk, H, W = self.cfg["tracking"], self.H, self.W
        while True:
            item = self._q_topk.get()
            if item is None:
                return
            plan, s, kind, i, draw = item
            t0 = time.perf_counter()
            if _DIAG_OFF == "topk":
                s._part_done()
                continue
            if kind == "track":
                s.track_idx.copy_(self._pixels_mix(self._lattice_track, plan.depth, self.n_track, draw))
            else:
                n_kf, n_cur = ba_ray_counts(self.cfg, plan.ba_kf_ids.shape[0])
                if tk["iter_RO"] == 0:                      # sample_valid_pixels_random
                    idx = hostrng.topk_valid_pixels(plan.depth, draw, n_cur)
                else:
                    idx = self._pixels_mix((tk["RO"]["n_rows"], tk["RO"]["n_cols"]), plan.depth, n_cur, draw)
                rows, owner, _ = s.ba_packed()
                rows[i, n_kf:n_kf + n_cur].copy_(idx + plan.kf_slot_base)
                owner[i, n_kf:n_kf + n_cur].fill_(-1)
            self.host_ms["topk"] += (time.perf_counter() - t0) * 1e3
            s._part_done()

    def _pixels_mix(self, lattice, depth, num, draw):
        """pixel_rc_to_indices(*sample_pixels_mix(H, W, *lattice, depth, num, draw)): the lattice pixels, then the top
        scores among the other valid pixels -- scores and top-k in one pass of hostrng.topk_valid_pixels (torch.topk's
        indices in torch.topk's order; 0.3-0.6 instead of 1.9 ms per call)"""
        if lattice not in self._lattice_masks:
            rows, cols = sh.sample_pixels_uniformly(self.H, self.W, lattice[0], lattice[1])
            idx = sh.pixel_rc_to_indices(rows, cols, self.H, self.W)
            mask = torch.zeros(self.H * self.W, dtype=torch.uint8)
            mask[idx] = 1
            self._lattice_masks[lattice] = (idx, mask)
        idx, mask = self._lattice_masks[lattice]
        return torch.cat([idx, hostrng.topk_valid_pixels(depth, draw, num - idx.numel(), mask)], 0)

    def _python_stream(self):
        """python's ``random`` generator: the keyframe-ray draws of every mapping iteration (keyframeSet.py:386-436),
        and the tracking pixels when iter_RO == 0 (select_samples, mipsfusion.py:510-515, its H-for-W quirk included)."""
        torch.set_num_threads(1)
        tk = self.cfg["tracking"]
        while True:
            item = self._q_py.get()
            if item is None:
                return
            plan, s = item
            t0 = time.perf_counter()
            if _DIAG_OFF == "python":
                s._part_done()
                continue
            # the draws go through hostrng's replica of random.sample(range(n), k) (same generator, same indices, checked
            # against python's at start-up, python's own otherwise): 25 -> 1 ms per mapping round of 15 iterations here
            with hostrng.py_session() as r:
                if plan.track and tk["iter_RO"] == 0:
                    iH, iW = tk["ignore_edge_H"], tk["ignore_edge_W"]
                    hh = self.H - 2 * iH
                    indice = r.sample_range(hh * (self.W - 2 * iW), int(self.n_track))      # sh.select_samples
                    ih, iw = torch.remainder(indice, hh), torch.div(indice, hh, rounding_mode="floor")
                    s.track_idx.copy_((ih + iH) * self.W + (iw + iW))
                if plan.ba_kf_ids is not None:
                    n_kf, _ = ba_ray_counts(self.cfg, plan.ba_kf_ids.shape[0])
                    first = plan.ba_kf_ids[0]
                    rows, owner, _ = s.ba_packed()
                    for i in range(self.it_ba):
                        self.gate.wait()
                        flat, _, kf_indices = self._index_db.indices_in_submap(first, plan.ba_kf_ids, n_kf, r.sample_range)
                        rows[i, :n_kf].copy_(flat)
                        owner[i, :n_kf].copy_(kf_indices)
            self.host_ms["python_rng"] += (time.perf_counter() - t0) * 1e3
            s._part_done()


def sequential_reference_samples(cfg, H, W, rays_per_kf, plans: List[FramePlan]):
    """The same index sets and jitter produced by calling the sampling functions one after the other in the
    reference's program order on the calling thread (what the unchanged caller does): the producer's checker."""
    tr, tk, mp = cfg["training"], cfg["tracking"], cfg["mapping"]
    S = tr["n_samples_d"] + tr["n_range_d"]
    db = DeviceRayDB.__new__(DeviceRayDB)
    db.num_rays_to_save = rays_per_kf
    lattice = (cfg["sampling"]["n_rays_h"], cfg["sampling"]["n_rays_w"]) if "sampling" in cfg else \
        (tk["RO"]["n_rows"], tk["RO"]["n_cols"])
    out = []
    for plan in plans:
        rec = {}
        if plan.track:
            if tk["iter_RO"] == 0:
                iH, iW = tk["ignore_edge_H"], tk["ignore_edge_W"]
                hh = H - 2 * iH
                indice = sh.select_samples(hh, W - 2 * iW, tk["sample"])
                ih, iw = torch.remainder(indice, hh), torch.div(indice, hh, rounding_mode="floor")
                rec["track_idx"] = (ih + iH) * W + (iw + iW)
            else:
                rows, cols = sh.sample_pixels_mix(H, W, lattice[0], lattice[1], plan.depth, tk["sample"])
                rec["track_idx"] = sh.pixel_rc_to_indices(rows, cols, H, W)
            rec["track_noise"] = torch.stack([torch.rand(tk["sample"], S) for _ in range(tk["iter"])])
        if plan.ba_kf_ids is not None:
            n_kf, n_cur = ba_ray_counts(cfg, plan.ba_kf_ids.shape[0])
            rows_l, own_l, noise_l = [], [], []
            for _ in range(mp["iters"]):
                flat, _, kf_indices = db.indices_in_submap(plan.ba_kf_ids[0], plan.ba_kf_ids, n_kf)
                if tk["iter_RO"] == 0:
                    idx = sh.sample_valid_pixels_random(plan.depth, n_cur)
                else:
                    r, c = sh.sample_pixels_mix(H, W, tk["RO"]["n_rows"], tk["RO"]["n_cols"], plan.depth, n_cur)
                    idx = sh.pixel_rc_to_indices(r, c, H, W)
                rows_l.append(torch.cat([flat, idx + plan.kf_slot_base]))
                own_l.append(torch.cat([kf_indices, -torch.ones(n_cur, dtype=torch.int64)]))
                noise_l.append(torch.rand(n_kf + n_cur, S))
            rec["ba_rows"], rec["ba_owner"], rec["ba_noise"] = torch.stack(rows_l), torch.stack(own_l), torch.stack(noise_l)
        out.append(rec)
    return out


# ======================================================================================================= the loop
def _matrix_to_quaternion_np(R):
    """geometry_helper.matrix_to_quaternion for ONE 3x3 in numpy fp32 scalars (same operations, same order)."""
    f32 = np.float32
    a, b, c, d, e, f, g, h, i = (f32(v) for v in R.reshape(9))
    one = f32(1.0)
    mag2 = [one + a + e + i, one + a - e - i, one - a + e - i, one - a - e + i]
    mag = [np.sqrt(m) if m > 0 else f32(0.0) for m in mag2]
    table = ((mag[0] * mag[0], h - f, c - g, d - b), (h - f, mag[1] * mag[1], d + b, c + g),
             (c - g, d + b, mag[2] * mag[2], f + h), (d - b, g + c, h + f, mag[3] * mag[3]))
    best = int(np.argmax(np.array(mag, dtype=np.float32)))
    den = f32(2.0) * max(mag[best], f32(0.1))
    q = np.array([t / den for t in table[best]], dtype=np.float32)
    return -q if q[0] < 0 else q


def _qt_to_matrix_np(qt):
    """geometry_helper.qt_to_transform_matrix for ONE (w, x, y, z, tx, ty, tz) in numpy fp32 scalars -> [4,4] fp32."""
    f32 = np.float32
    w, x, y, z = (f32(v) for v in qt[:4])
    k = f32(2.0) / (w * w + x * x + y * y + z * z)
    one = f32(1.0)
    T = np.eye(4, dtype=np.float32)
    T[0, :3] = (one - k * (y * y + z * z), k * (x * y - z * w), k * (x * z + y * w))
    T[1, :3] = (k * (x * y + z * w), one - k * (x * x + z * z), k * (y * z - x * w))
    T[2, :3] = (k * (x * z - y * w), k * (y * z + x * w), one - k * (x * x + y * y))
    T[:3, 3] = qt[4:7]
    return T


def frame_rays(frame):
    """[H*W, 7] rows [direction | rgb | depth] of one frame (mipsfusion.py:296-297)."""
    return torch.cat([frame["direction"], frame["rgb"], frame["depth"][..., None]], -1).reshape(-1, 7)


def submap_timeline(n_frames, kf_every, schedule):
    """frame -> (active sub-map, its keyframe slots) as they are when that frame is tracked and bundle-adjusted (the frame's
    own keyframe / switch is processed AFTER its BA, mipsfusion.py:681-712).  Keyframe slot = frame // kf_every; a frame
    with a ("new",) event is the first keyframe of the new sub-map, one with ("back", s) joins sub-map s after the
    refinement of local_BA_switch (mipsfusion.py:379-444)."""
    subs, active, tl = {0: [0]}, 0, {}
    for k in range(1, n_frames):
        tl[k] = (active, list(subs[active]))
        if k % kf_every == 0:
            ev, slot = schedule.get(k), k // kf_every
            if ev is None:
                subs[active].append(slot)
            elif ev[0] == "new":
                active = len(subs)
                subs[active] = [slot]
            else:
                active = ev[1]
                if active not in subs:
                    raise ValueError(f"schedule: sub-map {active} does not exist at frame {k}")
                subs[active].append(slot)
    return tl


class GraphedSequence:
    """Tracking + mapping of a frame sequence with the iterations replayed as hipGraphs
    (mipsfusion_amd.graph.GraphedSteps): RandomOptimizer rounds (one replay) -> one replay of the ``tracking.iter``
    pose-only iterations -> every ``map_every``-th frame one replay of the ``mapping.iters`` local-BA iterations.  Rays are
    gathered inside the graphs from ONE device table [keyframe database | current frame].

    Several sub-maps (BASELINE config 3; ``schedule``): the active process of the reference owns ONE model whose
    parameters are swapped at a switch (mipsfusion.py:607-658), and so does this class -- the captured graphs keep reading
    the same parameter tensors, a switch is a handful of device-to-device copies:
      ("new",)      active_submap_switch_new + initialize_new_localMLP (mipsfusion.py:637-652, 198-222): the active
                    sub-map's parameters are stored, ``recover_initial_param()``, a fresh map optimiser (cleared in place),
                    ``first_iters`` initialisation iterations on the keyframe's pixels as replays of one 25-iteration graph;
      ("back", s)   active_submap_switch + local_BA_switch (mipsfusion.py:608-634, 379-444): store, load sub-map s (the map
                    optimiser is NOT rebuilt -- the reference keeps its object), ``tracking.switch.map_num`` pose-only
                    iterations on rays of s's keyframes + the frame's pixels with the switch learning rates (one replay).
    What replaces host control plane here (out of scope): the switch decisions of Manager.py are the fixed
    ``schedule``; the sub-maps share one estimated world frame (the reference re-bases the pose to the new sub-map's first
    keyframe, mipsfusion.py:652 -- a rigid change of coordinates that moves no work; the miniature parity run of
    tests/seq_harness.py keeps the reference's local frames), PoseCorrector's ICP rectification is dropped."""

    INIT_INNER = 25          # iterations per replay of the sub-map initialisation graph (500 = 20 replays)

    def __init__(self, cfg, dev, frames, kf_every=15, sampler="reference", first_iters=None, stream=None,
                 lookahead=None, graph_ro=True, gate_producer=True, ro_precision=None, schedule=None, decoder_precision=None,
                 device_handover=None):
        """lookahead: how many frames the sample producer runs ahead of the GPU (default: ``map_every``, one whole
        mapping period -- a BA round needs ~40 ms of serial generator work, a frame without BA ~4 ms, so the work only
        evens out over a period; the reference's own DataLoader prefetches 8 frames, mipsfusion.py:672).
        first_iters: initialisation iterations of a sub-map (default: mapping.first_iters, 500 in the reference's configs).
        schedule: {frame: ("new",) | ("back", submap)} at keyframe frames (see the class docstring).
        decoder_precision: arithmetic of the model's decoder (default: JointEncoding's own default, "bf16x6" = the reference's
        fp32 arithmetic; "f16x3" = the fast mode).  ro_precision: arithmetic of the RandomOptimizer rounds (default: the
        model's; "f16" = the opt-in plain-f16 rounds of BASELINE config 5, pose within 1e-3 of the reference's)."""
        from .RandomOptimizer import RandomOptimizer
        from .graph import GraphedSteps, work_stream
        from .model import JointEncoding
        from .optim import FusedAdam
        from . import synth
        assert sampler in ("reference", "device")
        self.cfg, self.dev, self.frames, self.sampler, self.kf_every = cfg, dev, frames, sampler, kf_every
        self.graph_ro = graph_ro and cfg["tracking"]["iter_RO"] > 0
        # device_handover (default; MIPSF_SEQ_HOST_HANDOVER=1 or False = round 4's loop): a frame's three stages hand their
        # pose over ON THE DEVICE -- RandomOptimizer state -> tracking Parameters (ops.pose_handover), tracking Parameters ->
        # the BA round's current-frame slot (a 28-byte copy) -- and the host reads ONE pose back per frame, behind the last
        # stage.  Round 4 synchronised after every stage to carry the pose through host 4x4 algebra.  Same arithmetic,
        # bit-identical poses; measured on bench.py's 31-frame sequence: 7.48-7.62 -> 7.26-7.33 ms per frame.
        if device_handover is None:
            device_handover = not os.environ.get("MIPSF_SEQ_HOST_HANDOVER")
        self.device_handover = bool(device_handover) and self.graph_ro
        self.gate_producer = gate_producer
        self.stream = stream if stream is not None else work_stream(dev)
        self._Graphed = GraphedSteps
        tr, mp, tk = cfg["training"], cfg["mapping"], cfg["tracking"]
        tk["RO"].setdefault("initial_scaling_factor", 0.02)
        tk["RO"].setdefault("rescaling_factor", 0.5)
        tk.setdefault("ignore_edge_H", 20), tk.setdefault("ignore_edge_W", 20)
        self.S = tr["n_samples_d"] + tr["n_range_d"]
        self.H, self.W, fx, fy, cx, cy = synth.intrinsics_after_crop(cfg)
        H, W = self.H, self.W
        bb = torch.from_numpy(np.array(mp["bound"]))
        nf = torch.from_numpy(np.array(mp["localMLP_max_len"]))
        self.model = JointEncoding(cfg, bb, nf).to(dev).train()
        if decoder_precision is not None:
            self.model.decoder_precision = decoder_precision
        self.model.accumulate_param_grads_in_place = True
        # map_accum_step 1, map_wait_step 0 is what every shipped configuration of the reference uses (mipsfusion.py:330-335):
        # the captured loops run ONE backward per map step and the optimiser kernel clears the gradients (the shipped values);
        # other values are honoured by the BA round's graph (_ba_step_fn: the conditions depend on the iteration number only,
        # so they are recorded as they fall), and the scatter then ADDS into the table's gradient instead of storing slices
        if mp.get("map_accum_step", 1) < 1 or mp.get("map_wait_step", 0) < 0:
            raise ValueError("mapping.map_accum_step must be >= 1 and mapping.map_wait_step >= 0")
        self.plain_map_steps = mp.get("map_accum_step", 1) == 1 and mp.get("map_wait_step", 0) == 0
        self.model.grid_grad_is_zero_at_backward = self.plain_map_steps    # every map step is one backward + map_opt.step(zero_grad=True)
        # recover_initial_param() at a switch is a device-to-device copy (the reference's initial_dict sits where the model was built)
        self.model.initial_dict = {k: v.to(dev) for k, v in self.model.initial_dict.items()}
        # What a capture thread hands over per frame.  The camera's ray directions are the same image for every frame of a
        # sequence (datasets/dataset.py: one get_camera_rays per dataset): they are written into the current-frame rows of the
        # table ONCE and a frame then uploads its colours and depths only (4.6 MB instead of 8: 96 + 5 us instead of 149 us
        # on the work stream, measured with event pairs around 50 copies); frames with directions of their own are uploaded whole.
        d0 = frames[0]["direction"]
        self.shared_directions = all(f["direction"] is d0 or torch.equal(f["direction"], d0) for f in frames)
        if self.shared_directions:
            self.host_rays = [torch.cat([f["rgb"], f["depth"][..., None]], -1).reshape(-1, 4).pin_memory() for f in frames]
            self._dir_host = d0.reshape(-1, 3).contiguous()
        else:
            self.host_rays = [frame_rays(f).pin_memory() for f in frames]
        ds = types.SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, rays_d=frames[0]["direction"])
        self.ro = RandomOptimizer(cfg, types.SimpleNamespace(dataset=ds, device=dev))
        # particle rounds: the model's arithmetic unless the caller opts in to a faster one ("f16": BASELINE config 5 "fp16
        # decoder"; the tracked pose stays within 1e-3 of the reference's, test_random_optimizer_f16_rounds_track_the_reference_pose)
        self.ro.decoder_precision = self.model.decoder_precision if ro_precision is None else ro_precision
        kr, kc = sh.sample_pixels_uniformly(H, W, 100, 300 if W >= 300 else W // 2)  # 30 000 rays per keyframe
        self.kf_rows, self.kf_cols = kr.to(dev), kc.to(dev)
        self.R = kr.shape[0]
        self.Kmax = len(frames) // kf_every + 2
        self.table = torch.zeros(self.Kmax * self.R + H * W, 7, device=dev)
        self.db = DeviceRayDB(self.Kmax, self.R, dev, storage=self.table)
        self.cur = self.table[self.Kmax * self.R:]
        if self.shared_directions:
            self.cur[:, :3].copy_(self._dir_host)
            self._stage4 = torch.empty(H * W, 4, device=dev)
        self.n_kf = 0
        self.iters = mp["iters"]
        self.first_iters = mp.get("first_iters", 500) if first_iters is None else first_iters
        # ---- sub-maps: key-frame slots (rows of the ray database) of each, the stored parameters of the inactive ones
        self.schedule = dict(schedule or {})
        for k, ev in self.schedule.items():
            assert k % kf_every == 0 and 0 < k < len(frames) and ev[0] in ("new", "back"), "switches happen at keyframe frames"
        self.submaps = {0: {"kfs": [], "state": None}}
        self.active = 0
        self._timeline = self._make_timeline(len(frames))
        self.kf_qt = torch.zeros(self.Kmax, 7, device=dev)                   # (quaternion | translation) of every keyframe slot
        self.kf_qt[:, 0] = 1.0
        # ---- static state of the captured iterations
        k_related = [len(kfs) for _, kfs in self._timeline.values()] or [1]
        self.n_ba_max = max(sum(ba_ray_counts(cfg, k)) for k in range(1, max(k_related) + 1))
        self.ba_rot = torch.nn.Parameter(torch.tensor([[1., 0., 0., 0.]], device=dev).repeat(self.Kmax, 1))
        self.ba_trans = torch.nn.Parameter(torch.zeros(self.Kmax, 3, device=dev))       # slot j-1 = j-th keyframe of the ACTIVE sub-map, -1 = current
        self.fixed = torch.eye(4, device=dev)[None].clone()                  # its first keyframe (never optimised)
        self.ba_rows = torch.zeros(self.iters, self.n_ba_max, dtype=torch.int64, device=dev)
        self.ba_owner = torch.zeros(self.iters, self.n_ba_max, dtype=torch.int64, device=dev)
        self.ba_noise = torch.zeros(self.iters, self.n_ba_max, self.S, device=dev)
        self.map_opt = FusedAdam([{"params": self.model.decoder.parameters(), "weight_decay": 1e-6, "lr": mp["lr_decoder"]},
                                  {"params": self.model.embed_fn.parameters(), "eps": 1e-15, "lr": mp["lr_embed"]}],
                                 betas=(0.9, 0.99), capturable=True)
        self.ba_popt = FusedAdam([{"params": self.ba_rot, "lr": mp["lr_rot"]}, {"params": self.ba_trans, "lr": mp["lr_trans"]}],
                                 capturable=True)
        self.n_track = tk["sample"]
        self.go_rot = torch.nn.Parameter(torch.tensor([[1., 0., 0., 0.]], device=dev))
        self.go_trans = torch.nn.Parameter(torch.zeros(1, 3, device=dev))
        self.go_idx = torch.zeros(self.n_track, dtype=torch.int64, device=dev)
        self.go_own = torch.zeros(self.n_track, dtype=torch.int64, device=dev)
        self.go_noise = torch.zeros(tk["iter"], self.n_track, self.S, device=dev)
        self.go_popt = FusedAdam([{"params": self.go_rot, "lr": tk["lr_rot"]}, {"params": self.go_trans, "lr": tk["lr_trans"]}],
                                 capturable=True)
        # sub-map initialisation (first frame / ("new",)): mapping.sample pixels of the keyframe, its pose fixed
        self.n_init = mp["sample"]
        self.init_rows = torch.zeros(self.INIT_INNER, self.n_init, dtype=torch.int64, device=dev)
        self.init_owner = torch.zeros(self.n_init, dtype=torch.int64, device=dev)
        self.init_noise = torch.zeros(self.INIT_INNER, self.n_init, self.S, device=dev)
        self.init_graph = None
        # pose-only refinement after a switch back (local_BA_switch): every keyframe pose of the sub-map fixed, one optimisable pose
        sw = tk.get("switch", {"lr_rot": tk["lr_rot"], "lr_trans": tk["lr_trans"], "map_num": self.iters})
        self.sw_iters = sw["map_num"]
        self.sw_rot = torch.nn.Parameter(torch.tensor([[1., 0., 0., 0.]], device=dev))
        self.sw_trans = torch.nn.Parameter(torch.zeros(1, 3, device=dev))
        self.sw_popt = FusedAdam([{"params": self.sw_rot, "lr": sw["lr_rot"]}, {"params": self.sw_trans, "lr": sw["lr_trans"]}],
                                 capturable=True)
        self.sw_fixed = torch.eye(4, device=dev)[None].repeat(self.Kmax, 1, 1).contiguous()
        self.n_sw_max = mp["sample"] + max(mp["sample"] // 1, mp["sample"] // 5) if self.schedule else 1
        self.sw_rows = torch.zeros(self.sw_iters, self.n_sw_max, dtype=torch.int64, device=dev)
        self.sw_owner = torch.zeros(self.sw_iters, self.n_sw_max, dtype=torch.int64, device=dev)
        self.sw_noise = torch.zeros(self.sw_iters, self.n_sw_max, self.S if self.schedule else 1, device=dev)
        self.sw_graphs: Dict[int, object] = {}
        self.ba_graphs: Dict[int, object] = {}
        self.go_graph = None
        self.producer = None
        self.lookahead = max(1, mp["map_every"] if lookahead is None else lookahead)
        if sampler == "reference":
            self.producer = ReferenceSampleProducer(cfg, H, W, self.R, max(k_related), slots=self.lookahead + 2)
        self.capture_ms = 0.0
        self._qt_host = torch.zeros(16, 7, dtype=torch.float32)
        if torch.cuda.is_available():
            self._qt_host = self._qt_host.pin_memory()
        self._qt_next = 0
        self._qt_dev = torch.zeros(7, device=dev)               # (quaternion | translation) staging of the device hand-over
        self._score_bufs = {}
        self._lattice_dev = {}

    def _make_timeline(self, n_frames):
        return submap_timeline(n_frames, self.kf_every, self.schedule)

    # --------------------------------------------------------------------------- state kept out of warm-ups / switches
    def _optimizers(self):
        return (self.map_opt, self.ba_popt, self.go_popt, self.sw_popt)

    def _guarded(self, make_graph):
        """Capture a graph WITHOUT training on its warm-up: ``GraphedSteps`` runs the step function eagerly before it
        records (allocator and lazy-initialisation warm-up), i.e. real optimisation steps on whatever the static buffers
        hold.  Parameters, poses and every optimiser's state (moments, device step counters, host step counts) are
        snapshotted before and put back after, so the model that enters the next frame does not depend on how many graph
        shapes a sequence needs (a long sequence used to receive tens of extra mapping rounds before frame 1)."""
        with torch.no_grad():
            params = list(self.model.parameters()) + [self.ba_rot, self.ba_trans, self.go_rot, self.go_trans, self.sw_rot,
                                                      self.sw_trans]
            saved = [(p, p.detach().clone()) for p in params]
            opt_saved = []
            for opt in self._optimizers():
                st = {p: (s["step"], s["exp_avg"].clone(), s["exp_avg_sq"].clone()) for p, s in opt.state.items() if s}
                dev_saved = {gi: (a.clone(), b.clone()) for gi, (a, b) in opt._dev.items()}
                opt_saved.append((opt, st, dev_saved))
        t0 = time.perf_counter()
        g = make_graph()
        torch.cuda.synchronize()
        self.capture_ms += (time.perf_counter() - t0) * 1e3
        with torch.no_grad():
            for p, v in saved:
                p.copy_(v)
                if p.grad is not None:
                    p.grad.zero_()
            for opt, st, dev_saved in opt_saved:
                for p, s in opt.state.items():
                    if not s:
                        continue
                    if p in st:
                        s["step"] = st[p][0]
                        s["exp_avg"].copy_(st[p][1]), s["exp_avg_sq"].copy_(st[p][2])
                    else:                               # state created by the warm-up: a fresh optimiser has zeros
                        s["step"] = 0
                        s["exp_avg"].zero_(), s["exp_avg_sq"].zero_()
                for gi, (a, b) in opt._dev.items():
                    if gi in dev_saved:
                        a.copy_(dev_saved[gi][0]), b.copy_(dev_saved[gi][1])
                    else:
                        a.zero_(), b.zero_()
        torch.cuda.synchronize()
        return g

    # ------------------------------------------------------------------------------------- captured iterations
    def _ba_step_fn(self, n):
        tcfg, mp = self.cfg["training"], self.cfg["mapping"]

        rows, owner, noise = packed(self.ba_rows, n), packed(self.ba_owner, n), packed(self.ba_noise, n)

        accum, wait = mp.get("map_accum_step", 1), mp.get("map_wait_step", 0)
        zero_map_grads = self._zero_map_grads

        def step(k):
            if k == 0 and not self.plain_map_steps:
                zero_map_grads()                          # mipsfusion.py:285 (what an unfinished accumulation left behind)
            ret = self.model.forward_from_table(self.table, rows[k], self.ba_rot, self.ba_trans, self.fixed, owner[k], noise[k],
                                                accumulate_in_place=True)
            backward_from_one(get_loss_from_ret(ret, tcfg))
            if (k + 1) % accum == 0:                      # mipsfusion.py:330-335
                if (k + 1) > wait:
                    self.map_opt.step(zero_grad=True)
                else:
                    zero_map_grads()
            if (k + 1) % mp["pose_accum_step"] == 0:
                self.ba_popt.step(zero_grad=True)
        return step

    def _zero_map_grads(self):
        gs = [p.grad for p in self.model.parameters() if p.grad is not None]
        if gs:
            torch._foreach_zero_(gs)

    def _go_step(self, k):
        if k == 0:
            self.model.frozen_weights(True)          # the map is frozen for the frame's tracking iterations: one pack
        ret = self.model.forward_from_table(self.cur, self.go_idx, self.go_rot, self.go_trans, None, self.go_own,
                                            self.go_noise[k], EMD_w=0., accumulate_in_place=True)
        backward_from_one(get_loss_from_ret(ret, self.cfg["training"]))
        self.go_popt.step(zero_grad=True)
        if k == self.cfg["tracking"]["iter"] - 1:
            self.model.frozen_weights(False)

    def _ba_graph(self, n):
        g = self.ba_graphs.get(n)
        if g is None:
            g = self.ba_graphs[n] = self._guarded(lambda: self._Graphed(self._ba_step_fn(n), self.iters, warmup=1, stream=self.stream))
        return g

    def _init_step(self, k):
        """One initialisation iteration of a sub-map (first_frame_mapping / initialize_new_localMLP, mipsfusion.py:172-190,
        206-221): mapping.sample pixels of the keyframe, its pose fixed (owner 0), map Adam step."""
        if k == 0 and not self.plain_map_steps:
            self._zero_map_grads()      # map_optimizer.zero_grad() at the top of every initialisation iteration (mipsfusion.py:176,
            #                             207): a BA round whose iters is no multiple of map_accum_step leaves gradients behind;
            #                             iterations k > 0 find them cleared by the step in front of them
        ret = self.model.forward_from_table(self.table, self.init_rows[k], self.ba_rot.detach(), self.ba_trans.detach(),
                                            self.fixed, self.init_owner, self.init_noise[k])
        backward_from_one(get_loss_from_ret(ret, self.cfg["training"]))
        self.map_opt.step(zero_grad=True)

    def _sw_step_fn(self, n):
        """local_BA_switch (mipsfusion.py:379-444) for n rays: the keyframe poses of the sub-map are all fixed, the overlapping
        frame's pose is the one optimisable pose (switch learning rates), stepped every pose_accum_step iterations; the map
        is not stepped (the reference lets its gradients pile up unused until the next local BA clears them: the graph is
        recorded with the map frozen, which gives the same poses without computing them)."""
        rows, owner, noise = packed(self.sw_rows, n), packed(self.sw_owner, n), packed(self.sw_noise, n)
        accum = self.cfg["mapping"]["pose_accum_step"]

        def step(k):
            if k == 0:
                self.model.frozen_weights(True)
            ret = self.model.forward_from_table(self.table, rows[k], self.sw_rot, self.sw_trans, self.sw_fixed, owner[k], noise[k],
                                                accumulate_in_place=True)
            backward_from_one(get_loss_from_ret(ret, self.cfg["training"]))
            if (k + 1) % accum == 0:
                self.sw_popt.step(zero_grad=True)
            if k == self.sw_iters - 1:
                self.model.frozen_weights(False)
        return step

    def _frozen_map(self, make_graph):
        for prm in self.model.parameters():
            prm.requires_grad_(False)
        try:
            return self._guarded(make_graph)
        finally:
            for prm in self.model.parameters():
                prm.requires_grad_(True)

    def _sw_graph(self, n):
        g = self.sw_graphs.get(n)
        if g is None:
            g = self.sw_graphs[n] = self._frozen_map(lambda: self._Graphed(self._sw_step_fn(n), self.sw_iters, warmup=1, stream=self.stream))
        return g

    # --------------------------------------------------------------------------------------------- device sampler
    def _scratch(self, rows, n):
        """persistent score buffer [rows, n]: a fresh 17 MB temporary (three of them) per BA round made the caching
        allocator release and re-request device memory -- milliseconds of host stall per round"""
        buf = self._score_bufs.get((rows, n))
        if buf is None:
            buf = self._score_bufs[(rows, n)] = torch.empty(rows, n, dtype=torch.float32, device=self.dev)
        return buf

    def _device_draw(self, population, k, rows=1):
        """k distinct integers of range(population) per row, uniformly: top-k of i.i.d. uniforms (no replacement)."""
        return self._scratch(rows, population).uniform_().topk(k, dim=1).indices

    def _device_valid_pixels(self, k, rows, lattice=None):
        """sample_pixels_mix's distribution on the device: `rows` independent draws of k distinct VALID-depth pixels
        (|N(0,1)| scores, invalid and lattice pixels score 0: sampling_helper.py:28-32, 53-68)."""
        valid = (self.cur[:, 6] > 0).float()
        if lattice is not None:
            valid[lattice] = 0
        scores = self._scratch(rows, valid.shape[0]).normal_().abs_().mul_(valid[None])
        return scores.topk(k, dim=1).indices

    def _lattice(self, n_rows, n_cols):
        """flat pixel indices of the uniform lattice (sample_pixels_uniformly), on the device, built once"""
        lat = self._lattice_dev.get((n_rows, n_cols))
        if lat is None:
            rows, cols = sh.sample_pixels_uniformly(self.H, self.W, n_rows, n_cols)
            lat = self._lattice_dev[(n_rows, n_cols)] = (rows * self.W + cols).to(self.dev)
        return lat

    def _slots_dev(self, slots):
        key = tuple(slots)
        if getattr(self, "_slots_key", None) != key:
            self._slots_key, self._slots_t = key, torch.tensor(list(slots), dtype=torch.int64, device=self.dev)
        return self._slots_t

    def _fill_ba_device(self, slots):
        """Device draws with the per-keyframe shares of sample_rays_in_submap (keyframeSet.py:386-436) over the keyframe
        slots `slots` of the active sub-map + sample_pixels_mix's distribution for the current frame."""
        K = len(slots)
        n_kf, n_cur = ba_ray_counts(self.cfg, K)
        R, it = self.R, self.iters
        idx, own = [], []
        n_first = max(n_kf // K, n_kf // 10)
        idx.append(self._device_draw(R, n_first, it) + slots[0] * R)
        own.append(torch.zeros(it, n_first, dtype=torch.int64, device=self.dev))
        n_last = max(n_kf // K, n_kf // 5) if K > 2 else 0
        n_other = n_kf - n_first - n_last
        if K > 1 and n_other:
            span = (K - 2) if K > 2 else 1
            o = self._device_draw(span * R, n_other, it)
            pos = o // R + 1
            idx.append(self._slots_dev(slots)[pos] * R + (o - (pos - 1) * R)), own.append(pos)
        elif n_other:                       # one keyframe: every keyframe ray comes from it
            idx.append(self._device_draw(R, n_other, it) + slots[0] * R)
            own.append(torch.zeros(it, n_other, dtype=torch.int64, device=self.dev))
        if n_last:
            idx.append(self._device_draw(R, n_last, it) + slots[K - 1] * R)
            own.append(torch.full((it, n_last), K - 1, dtype=torch.int64, device=self.dev))
        tk = self.cfg["tracking"]
        lat = self._lattice(tk["RO"]["n_rows"], tk["RO"]["n_cols"])
        extra = self._device_valid_pixels(n_cur - lat.shape[0], it, lattice=lat)
        idx.append(torch.cat([lat[None].expand(it, -1), extra], 1) + self.Kmax * R)
        own.append(torch.full((it, n_cur), -1, dtype=torch.int64, device=self.dev))
        n = n_kf + n_cur
        packed(self.ba_rows, n).copy_(torch.cat(idx, 1)), packed(self.ba_owner, n).copy_(torch.cat(own, 1))
        packed(self.ba_noise, n).uniform_()
        return n

    def _fill_init_device(self):
        """select_samples (mipsfusion.py:135-138, 175-177): mapping.sample pixels of the keyframe, uniform over ALL pixels
        (drawn with replacement here: ~6 repeats among 1800 of 285 200), and the jitter, for INIT_INNER iterations."""
        torch.randint(0, self.H * self.W, tuple(self.init_rows.shape), out=self.init_rows)
        self.init_rows.add_(self.Kmax * self.R)
        self.init_noise.uniform_()

    def _sw_counts(self, K):
        mp = self.cfg["mapping"]
        return mp["sample"], max(mp["sample"] // K, mp["sample"] // 5)

    def _fill_sw_device(self, slots):
        """sample_rays_in_given_kf (keyframeSet.py:444-455) over the sub-map's keyframes + random pixels of the frame
        (mipsfusion.py:410-416), drawn on the device for the map_num iterations of one local_BA_switch."""
        K = len(slots)
        n_kf, n_ov = self._sw_counts(K)
        it, R = self.sw_iters, self.R
        o = torch.randint(0, K * R, (it, n_kf), device=self.dev)
        pos = o // R
        rows_kf = self._slots_dev(slots)[pos] * R + (o - pos * R)
        rows_ov = torch.randint(0, self.H * self.W, (it, n_ov), device=self.dev) + self.Kmax * R
        n = n_kf + n_ov
        packed(self.sw_rows, n).copy_(torch.cat([rows_kf, rows_ov], 1))
        packed(self.sw_owner, n).copy_(torch.cat([pos, torch.full((it, n_ov), -1, dtype=torch.int64, device=self.dev)], 1))
        packed(self.sw_noise, n).uniform_()
        return n

    def _fill_go_device(self):
        s = self.cfg.get("sampling", {"n_rays_h": self.cfg["tracking"]["RO"]["n_rows"], "n_rays_w": self.cfg["tracking"]["RO"]["n_cols"]})
        lat = self._lattice(s["n_rays_h"], s["n_rays_w"])
        extra = self._device_valid_pixels(self.n_track - lat.shape[0], 1, lattice=lat)[0]
        self.go_idx.copy_(torch.cat([lat, extra]))
        self.go_noise.uniform_()

    # ----------------------------------------------------------------------------------------------- bookkeeping
    def _set_pose(self, rot, trans, slot, pose):
        """pose: CPU [4,4].  The 4x4 <-> (quaternion, translation) conversions of the loop run on the HOST (a dozen tiny
        device launches each otherwise); two 16-byte uploads hand the result over."""
        # ... in numpy fp32 scalars, not torch: every torch call drops and re-takes the GIL, and while the sample
        # producer's python stage works each re-take queues behind it (30 tiny torch ops: 0.15 -> 5 ms, see
        # ReferenceSampleProducer).  Same operations in the same order as geometry_helper.matrix_to_quaternion.
        with torch.no_grad():
            R = pose.detach().to("cpu", torch.float32).numpy()
            self._qt_next = (self._qt_next + 1) % self._qt_host.shape[0]      # ring of pinned staging rows: no wait
            stage = self._qt_host[self._qt_next]
            h = stage.numpy()
            h[:4] = _matrix_to_quaternion_np(R[:3, :3])
            h[4:] = R[:3, 3]
            rot[slot].copy_(stage[:4], non_blocking=True)
            trans[slot].copy_(stage[4:], non_blocking=True)

    def _get_pose(self, rot, trans, slot):
        """-> CPU [4,4] of the optimised (quaternion, translation) in `slot` (one 28-byte read-back)."""
        qt = torch.cat([rot.detach()[slot], trans.detach()[slot]]).cpu().numpy()
        return torch.from_numpy(_qt_to_matrix_np(qt))

    def _store_keyframe_rays(self, slot):
        self.db.store(slot, self.cur.view(self.H, self.W, 7)[self.kf_rows, self.kf_cols])

    def _add_keyframe(self, pose):
        """the current frame becomes the next keyframe of the ACTIVE sub-map (its pose: position j of the pose slots)"""
        slot, kfs = self.n_kf, self.submaps[self.active]["kfs"]
        self._store_keyframe_rays(slot)
        self._set_pose(self.kf_qt[:, :4], self.kf_qt[:, 4:], slot, pose)
        if not kfs:
            self.fixed[0].copy_(pose.to(self.dev))
        else:
            self._set_pose(self.ba_rot, self.ba_trans, len(kfs) - 1, pose)
        kfs.append(slot)
        self.n_kf += 1

    def _state_tensors(self):
        return [p.data for p in self.model.parameters()]

    def _store_active(self):
        """the active sub-map leaves the model: parameters -> its store (device-to-device; the reference copies them to the
        other process through shared memory, mipsfusion.py:616, 642, InactiveMap.py:66-70), optimised keyframe poses ->
        the keyframe table"""
        sm = self.submaps[self.active]
        with torch.no_grad():
            if sm["state"] is None:
                sm["state"] = [t.clone() for t in self._state_tensors()]
            else:
                for dst, src in zip(sm["state"], self._state_tensors()):
                    dst.copy_(src)
            k = len(sm["kfs"])
            if k > 1:
                self.kf_qt[self._slots_dev(sm["kfs"])[1:]] = torch.cat([self.ba_rot.detach()[:k - 1], self.ba_trans.detach()[:k - 1]], 1)

    def _switch_new(self, pose):
        """active_submap_switch_new + initialize_new_localMLP (mipsfusion.py:637-652, 198-222)"""
        self._store_active()
        self.model.recover_initial_param()
        self.map_opt.reset()                                # create_optimizer(): a fresh Adam
        self.active = len(self.submaps)
        self.submaps[self.active] = {"kfs": [], "state": None}
        self._add_keyframe(pose)                            # first keyframe of the new sub-map: fixed[0]
        self._initialise()

    def _initialise(self):
        if self.init_graph is None:
            self._fill_init_device()
            self.init_graph = self._guarded(lambda: self._Graphed(self._init_step, self.INIT_INNER, warmup=1, stream=self.stream))
        for _ in range(max(1, self.first_iters // self.INIT_INNER)):
            self._fill_init_device()
            self.init_graph.replay()

    def _switch_back(self, target, pose, waiting):
        """active_submap_switch + local_BA_switch (mipsfusion.py:608-634, 379-444) -> the refined pose of the frame (CPU)"""
        self._store_active()
        sm = self.submaps[target]
        with torch.no_grad():
            for dst, src in zip(self._state_tensors(), sm["state"]):
                dst.copy_(src)                              # load_state_dict of the asked sub-map, device to device
            self.active = target
            slots = sm["kfs"]
            k = len(slots)
            qt = self.kf_qt[self._slots_dev(slots)]
            poses = qt_to_transform_matrix(qt[:, :4], qt[:, 4:])
            self.sw_fixed[:k].copy_(poses)
            self.fixed[0].copy_(poses[0])
            if k > 1:
                self.ba_rot.data[:k - 1].copy_(qt[1:, :4]), self.ba_trans.data[:k - 1].copy_(qt[1:, 4:])
        self._set_pose(self.sw_rot, self.sw_trans, 0, pose)
        self.sw_popt.reset()
        n = self._fill_sw_device(slots)
        self._sw_graph(n).replay()
        pose = waiting(lambda: self._get_pose(self.sw_rot, self.sw_trans, 0))
        self._add_keyframe(pose)                            # the overlapping keyframe joins the sub-map switched to
        return pose

    def _n_kf_at(self, k):
        """related keyframes when frame k's local BA runs (the frame itself is added AFTER its BA, mipsfusion.py:681-688)"""
        return len(self._timeline[k][1])

    def _plan(self, k):
        ba = torch.tensor(self._timeline[k][1]) if (k % self.cfg["mapping"]["map_every"] == 0) else None
        return FramePlan(k, self.frames[k]["depth"], True, ba, self.Kmax * self.R)

    def _load_ba(self, s: FrameSamples):
        n = s.n_ba
        for dst, src in zip((self.ba_rows, self.ba_owner, self.ba_noise), s.ba_packed()):
            packed(dst, n).copy_(src, non_blocking=True)                     # contiguous on both sides: plain DMA
        return n

    def _upload_frame(self, k):
        """Frame k's pixels into the current-frame rows of the device table, on the work stream (pinned source)."""
        if self.shared_directions:
            self._stage4.copy_(self.host_rays[k], non_blocking=True)
            self.cur[:, 3:7].copy_(self._stage4)
        else:
            self.cur.copy_(self.host_rays[k], non_blocking=True)

    # ------------------------------------------------------------------------------------------------------ run
    def first_frame(self, gt_pose):
        """mipsfusion.py:155-194: ground-truth pose, ``first_iters`` mapping iterations on frame 0's pixels."""
        self._upload_frame(0)
        pose0 = gt_pose.float().cpu()
        self._add_keyframe(pose0)
        self._initialise()
        self._set_pose(self.go_rot, self.go_trans, 0, pose0)
        self._fill_go_device()
        # the tracking graph is recorded with the map frozen
        self.go_graph = self._frozen_map(lambda: self._Graphed(self._go_step, self.cfg["tracking"]["iter"], warmup=1, stream=self.stream))
        if self.graph_ro:
            t0 = time.perf_counter()
            self.ro.capture(self.model, self.cfg["tracking"]["iter_RO"], self.stream)
            torch.cuda.synchronize()
            self.capture_ms += (time.perf_counter() - t0) * 1e3
        # one-off lazy initialisations (the first torch._foreach_zero_ takes 65 ms) belong to the set-up, not to frame 1
        self.go_popt.reset(), self.ba_popt.reset(), self.sw_popt.reset()
        torch.cuda.synchronize()
        return pose0

    def precapture(self):
        """Capture every graph shape the sequence will need up front (one-off cost, reported as `capture_ms`); every
        capture leaves parameters and optimiser state as it found them (`_guarded`)."""
        mp = self.cfg["mapping"]
        n_frames = len(self.frames)
        slot_sets = {}
        for k in range(1, n_frames):
            if k % mp["map_every"] == 0:
                slot_sets.setdefault(len(self._timeline[k][1]), self._timeline[k][1])
        for K, slots in sorted(slot_sets.items()):
            self._ba_graph(sum(ba_ray_counts(self.cfg, K)))
        if self.producer is None:
            # the device sampler's draws have shapes that depend on K: torch's first top-k / index op of a new shape
            # costs 10-70 ms (measured: 72 ms in the first BA round after a keyframe was added)
            for K, slots in sorted(slot_sets.items()):
                self._fill_ba_device(slots)
        for k, ev in sorted(self.schedule.items()):
            if ev[0] == "back":                         # the sub-map switched to has these keyframes at that moment
                subs = {}
                for kk in range(1, k + 1):
                    subs[self._timeline[kk][0]] = self._timeline[kk][1]
                slots = subs[ev[1]]
                self._sw_graph(sum(self._sw_counts(len(slots))))
                self._fill_sw_device(slots)
                with torch.no_grad():                   # the pose-table operations of a switch, once per shape (torch loads a
                    qt = self.kf_qt[self._slots_dev(slots)]      # kernel of a new shape lazily: 10-70 ms inside the switch frame)
                    qt_to_transform_matrix(qt[:, :4], qt[:, 4:])
                    self.kf_qt[self._slots_dev(slots)[1:]] = qt[1:].clone()
                    torch.cat([self.ba_rot.detach()[:len(slots) - 1], self.ba_trans.detach()[:len(slots) - 1]], 1)
        torch.cuda.synchronize()

    def _keyframe_event(self, k, pose, waiting, t_switch, t3, last_switch, est):
        """mipsfusion.py:686-712, after the BA: the frame becomes a keyframe, or the schedule switches the sub-map"""
        ev = self.schedule.get(k)
        if ev is None:
            self._add_keyframe(pose)
        elif ev[0] == "new":
            self._switch_new(pose)
            last_switch = k
        else:
            pose = self._switch_back(ev[1], pose, waiting)
            last_switch = k
        if ev is not None:
            waiting(torch.cuda.synchronize)
            t_switch[k] = {"kind": ev[0], "ms": round((time.perf_counter() - t3) * 1e3, 3)}
        est.append(pose)
        return last_switch

    def _frame_device_handover(self, k, est, last_switch, gate, waiting, staged, t_frame, t_ro, t_go, t_ba, t_wait, detail):
        """One frame with the pose handed over on the device (see __init__): everything is enqueued, ONE read-back at the end.
        -> (pose [4,4] CPU, None)."""
        cfg, mp = self.cfg, self.cfg["mapping"]
        n_frames = len(self.frames)
        t0 = time.perf_counter()
        if gate is not None:
            gate.clear()
        # (the next frame's 8 MB staged ahead on a copy stream and moved device to device here was measured: the steady frames
        # do not change and every few frames one takes 8-20 ms -- the upload stays on the work stream)
        self._upload_frame(k)
        samples, wait_ms = None, 0.0
        if self.producer is not None and k + self.lookahead < n_frames:
            self.producer.submit(self._plan(k + self.lookahead))
        prev = est[-1]
        if len(est) < 2 or (k - last_switch) < 2:                           # predict_current_pose (mipsfusion.py:448-457)
            init = prev
        else:
            init = torch.from_numpy(prev.numpy() @ np.linalg.inv(est[-2].numpy()) @ prev.numpy())
        if self.producer is not None:
            tw = time.perf_counter()
            samples = self.producer.get()
            wait_ms = (time.perf_counter() - tw) * 1e3
            assert samples.frame_id == k
            self.go_idx.copy_(samples.track_idx, non_blocking=True)
            self.go_noise.copy_(samples.track_noise, non_blocking=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        self.ro.enqueue_graphed(self.cur[:, 6], init)
        ops.pose_handover(self.ro.tracked_pose_dev, self.go_rot.detach(), self.go_trans.detach())
        ev[1].record()
        t1 = time.perf_counter()
        self.go_popt.reset()
        if self.producer is None:
            self._fill_go_device()
        t1b = time.perf_counter()
        self.go_graph.replay()
        ev[2].record()
        do_ba = k % mp["map_every"] == 0
        t2 = time.perf_counter()
        if do_ba:
            n = (self._load_ba(samples) if samples is not None else self._fill_ba_device(self.submaps[self.active]["kfs"]))
            slots = self.submaps[self.active]["kfs"]
            assert slots == self._timeline[k][1], "the sub-map bookkeeping left the planned timeline"
            # the tracked pose -> the round's current-frame slot, through the 4x4 the reference passes (a unit quaternion again)
            torch.cat([self.go_rot.detach()[0], self.go_trans.detach()[0]], out=self._qt_dev)
            ops.pose_handover(self._qt_dev, self.ba_rot.detach()[-1], self.ba_trans.detach()[-1], quaternion=True)
            self.ba_popt.reset()
            t2b = time.perf_counter()
            self._ba_graph(n).replay()
            ev[3].record()
            pose = waiting(lambda: self._get_pose(self.ba_rot, self.ba_trans, -1))
        else:
            pose = waiting(lambda: self._get_pose(self.go_rot, self.go_trans, 0))      # the read-back synchronises
        t3 = time.perf_counter()
        if samples is not None:
            self.producer.release(samples)
        ro_ms, go_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
        ba_ms = ev[2].elapsed_time(ev[3]) if do_ba else 0.0
        detail["go_fill_ms"].append((t1b - t1) * 1e3), detail["go_replay_ms"].append(go_ms)
        detail["go_launch_ms"].append((t2 - t1b) * 1e3), detail["go_gpu_ms"].append(go_ms)
        if do_ba:
            detail["ba_fill_ms"].append((t2b - t2) * 1e3), detail["ba_replay_ms"].append(ba_ms)
        detail["tail_ms"].append(0.0)
        t_frame.append((t3 - t0) * 1e3), t_ro.append(ro_ms), t_go.append(go_ms), t_ba.append(ba_ms), t_wait.append(wait_ms)
        return pose, staged

    def run(self, gt_poses, precapture=True):
        """-> dict of per-frame wall-clock lists (ms) and the estimated poses."""
        cfg, dev = self.cfg, self.dev
        tk, mp = cfg["tracking"], cfg["mapping"]
        n_frames = len(self.frames)
        est = [self.first_frame(gt_poses[0])]
        if precapture:
            self.precapture()
        if self.producer is not None:
            for j in range(1, min(n_frames, 1 + self.lookahead)):
                self.producer.submit(self._plan(j))
        t_frame, t_ro, t_go, t_ba, t_wait, t_switch = [], [], [], [], [], {}
        detail = {"go_fill_ms": [], "go_replay_ms": [], "go_launch_ms": [], "go_gpu_ms": [], "ba_fill_ms": [], "ba_replay_ms": [], "tail_ms": []}
        gate = self.producer.gate if (self.producer is not None and self.gate_producer) else None
        last_switch = 0

        def waiting(fn):
            """fn blocks on the GPU: the producer stages may use the host meanwhile"""
            if gate is None:
                return fn()
            gate.set()
            try:
                return fn()
            finally:
                gate.clear()

        staged = None
        for k in range(1, n_frames):
            if self.device_handover:
                pose, staged = self._frame_device_handover(k, est, last_switch, gate, waiting, staged, t_frame, t_ro, t_go, t_ba,
                                                           t_wait, detail)
                t3 = time.perf_counter()
                if k % self.kf_every == 0:
                    last_switch = self._keyframe_event(k, pose, waiting, t_switch, t3, last_switch, est)
                else:
                    est.append(pose)
                continue
            t0 = time.perf_counter()
            if gate is not None:
                gate.clear()
            self._upload_frame(k)
            samples, wait_ms = None, 0.0
            if self.producer is not None:
                if k + self.lookahead < n_frames:
                    self.producer.submit(self._plan(k + self.lookahead))     # `lookahead` frames ahead of the GPU
            prev = est[-1]                                                   # poses live on the host (4x4 algebra there)
            if len(est) < 2 or (k - last_switch) < 2:                        # predict_current_pose (mipsfusion.py:448-457)
                init = prev
            else:                                                            # constant velocity, in numpy (see _set_pose)
                init = torch.from_numpy(prev.numpy() @ np.linalg.inv(est[-2].numpy()) @ prev.numpy())
            if self.producer is not None:
                # the frame's host-drawn samples go up BEFORE the RandomOptimizer rounds (they depend on nothing the GPU
                # computes: the uploads run while the host would otherwise sit in the rounds' read-back)
                tw = time.perf_counter()
                samples = self.producer.get()
                wait_ms = (time.perf_counter() - tw) * 1e3
                assert samples.frame_id == k
                self.go_idx.copy_(samples.track_idx, non_blocking=True)
                self.go_noise.copy_(samples.track_noise, non_blocking=True)
            if self.graph_ro:
                pose = self.ro.optimize_graphed(self.cur[:, 6], init, waiting=waiting)
            else:
                self.model.eval()
                pose = self.ro.optimize(self.model, self.cur.view(self.H, self.W, 7)[..., 6], init.to(dev), None,
                                        n_iter=tk["iter_RO"]).cpu()
                self.model.train()
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            self._set_pose(self.go_rot, self.go_trans, 0, pose)
            ta = time.perf_counter()
            self.go_popt.reset()
            if self.producer is None:
                self._fill_go_device()
            else:
                detail.setdefault("go_fill_parts_ms", []).append([round((ta - t1) * 1e3, 2), round((time.perf_counter() - ta) * 1e3, 2)])
            t1b = time.perf_counter()
            ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev_a.record()
            self.go_graph.replay()
            ev_b.record()
            t1c = time.perf_counter()
            n_ba_loaded = None
            if k % mp["map_every"] == 0:                    # the BA batch is queued behind the tracking graph: the host would be idle
                n_ba_loaded = (self._load_ba(samples) if samples is not None      # in the read-back (same draws in the same order)
                               else self._fill_ba_device(self.submaps[self.active]["kfs"]))
            pose = waiting(lambda: self._get_pose(self.go_rot, self.go_trans, 0))     # the read-back synchronises
            t2 = time.perf_counter()
            detail["go_fill_ms"].append((t1b - t1) * 1e3), detail["go_replay_ms"].append((t2 - t1b) * 1e3)
            detail["go_launch_ms"].append((t1c - t1b) * 1e3)
            detail["go_gpu_ms"].append(ev_a.elapsed_time(ev_b))
            ba_ms = 0.0
            if k % mp["map_every"] == 0:
                slots = self.submaps[self.active]["kfs"]
                assert slots == self._timeline[k][1], "the sub-map bookkeeping left the planned timeline"
                self._set_pose(self.ba_rot, self.ba_trans, -1, pose)
                self.ba_popt.reset()
                n = n_ba_loaded
                t2b = time.perf_counter()
                self._ba_graph(n).replay()
                pose = waiting(lambda: self._get_pose(self.ba_rot, self.ba_trans, -1))
                ba_ms = (time.perf_counter() - t2) * 1e3
                detail["ba_fill_ms"].append((t2b - t2) * 1e3), detail["ba_replay_ms"].append(ba_ms - (t2b - t2) * 1e3)
            t3 = time.perf_counter()
            if k % self.kf_every == 0:                                       # mipsfusion.py:686-712, after the BA
                ev = self.schedule.get(k)
                if ev is None:
                    self._add_keyframe(pose)
                elif ev[0] == "new":
                    self._switch_new(pose)
                    last_switch = k
                else:
                    pose = self._switch_back(ev[1], pose, waiting)
                    last_switch = k
                if ev is not None:
                    waiting(torch.cuda.synchronize)
                    t_switch[k] = {"kind": ev[0], "ms": round((time.perf_counter() - t3) * 1e3, 3)}
            est.append(pose)
            waiting(torch.cuda.synchronize)
            if samples is not None:
                self.producer.release(samples)
            detail["tail_ms"].append((time.perf_counter() - t3) * 1e3)
            t_frame.append((time.perf_counter() - t0) * 1e3), t_ro.append((t1 - t0) * 1e3)
            t_go.append((t2 - t1) * 1e3), t_ba.append(ba_ms), t_wait.append(wait_ms)
        if self.producer is not None:
            self.producer.gate.set()
            self.producer.close()
        return {"frame_ms": t_frame, "ro_ms": t_ro, "go_ms": t_go, "ba_ms": t_ba, "producer_wait_ms": t_wait,
                "est": est, "capture_ms": self.capture_ms, "detail_ms": detail, "switch": t_switch,
                "submaps": {s: list(v["kfs"]) for s, v in self.submaps.items()},
                "producer_host_ms": dict(self.producer.host_ms) if self.producer is not None else None}


def summarise(res, gt_poses, cfg, launch):
    fm = np.array(res["frame_ms"])
    err = [float((res["est"][k][:3, 3].cpu().float() - gt_poses[k][:3, 3].float()).norm()) for k in range(len(res["est"]))]
    ba = [t for t in res["ba_ms"] if t > 0]
    out = {"frames": len(res["est"]), "frames_timed": len(fm),
           "frames_timed_note": "frame 0 (its mapping.first_iters initialisation iterations, graph captures) is set-up and not "
                                "among the timed frames; every later frame counts",
           "ms_per_frame_mean": round(float(fm.mean()), 3),
           "ms_per_frame_median": round(float(np.median(fm)), 3), "ms_per_frame_p95": round(float(np.percentile(fm, 95)), 3),
           "ro_ms_mean": round(float(np.mean(res["ro_ms"])), 3), "go_ms_mean": round(float(np.mean(res["go_ms"])), 3),
           "ba_ms_per_round_median": round(float(np.median(ba)), 3) if ba else None,
           "producer_wait_ms_mean": round(float(np.mean(res["producer_wait_ms"])), 3),
           "graph_capture_ms_one_off": round(res["capture_ms"], 1),
           "cadence": {"iter_RO": cfg["tracking"]["iter_RO"], "tracking_iter": cfg["tracking"]["iter"],
                       "mapping_iters": cfg["mapping"]["iters"], "map_every": cfg["mapping"]["map_every"]},
           "launch": launch, "ate_rmse_m": round(float(np.sqrt(np.mean(np.square(err)))), 4),
           "ate_max_m": round(max(err), 4)}
    if res.get("switch"):
        sw = {int(k): v for k, v in res["switch"].items()}
        plain = np.array([t for i, t in enumerate(fm) if (i + 1) not in sw])
        out["switch_frames"] = sw
        out["submap_keyframe_slots"] = res.get("submaps")
        out["ms_per_frame_mean_without_switch_frames"] = round(float(plain.mean()), 3)
        out["switch_frames_note"] = ("ms_per_frame_mean / median / p95 include the switch frames: 'new' = parameter store + "
                                     "recover_initial_param + fresh map optimiser + mapping.first_iters initialisation iterations, "
                                     "'back' = store + load + tracking.switch.map_num pose-only iterations")
    if res.get("detail_ms"):
        out["detail_ms_mean"] = {k: round(float(np.mean(v)), 3) for k, v in res["detail_ms"].items() if len(v) and k != "go_fill_parts_ms"}
        out["frame_ms_all"] = [round(float(t), 2) for t in fm]
        if os.environ.get("MIPSF_SEQ_VERBOSE"):
            out["ro_go_ms_all"] = [[round(float(a), 2), round(float(b), 2)] for a, b in zip(res["ro_ms"], res["go_ms"])]
            out["go_fill_parts_ms_all"] = res["detail_ms"].get("go_fill_parts_ms")
    if res.get("producer_host_ms"):
        n = max(1, len(res["frame_ms"]))
        out["producer_host_ms_per_frame"] = {k: round(v / n, 3) for k, v in res["producer_host_ms"].items()}
    return out
