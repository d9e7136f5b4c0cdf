"""Tensor-level wrappers of the C ABI and the autograd Functions built on them.

Everything here launches HIP kernels on the caller's current stream through ``_lib``; outputs are
torch-allocated and handed over by pointer (the kernels never allocate).  No CPU path exists.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import FEAT_AOS, FEAT_LEVEL_MAJOR, check, dptr, lib, stream_ptr

DECODER_PARAM_ORDER = ("pts_linear.0.weight", "pts_linear.0.bias", "pts_linear.2.weight", "pts_linear.2.bias",
                       "rgb_linear.0.weight", "rgb_linear.0.bias", "sdf_linear.0.weight", "sdf_linear.0.bias",
                       "sdf_linear.2.weight", "sdf_linear.2.bias")
_DEC_FIELDS = ("w_pts0", "b_pts0", "w_pts2", "b_pts2", "w_rgb0", "b_rgb0", "w_sdf0", "b_sdf0", "w_sdf2", "b_sdf2")
_DEC_SHAPES = ((128, 51), (128,), (128, 128), (128,), (3, 115), (3,), (128, 96), (128,), (5, 128), (5,))


# Optional per-kernel timing: set PROFILE = {} and every launch below is bracketed by a pair of events recorded
# on the launch stream (torch's current stream IS the stream the kernels are enqueued on).
PROFILE = None


class _timed:
    def __init__(self, name, stream=None):
        self.name, self.stream = name, stream

    def __enter__(self):
        if PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(self.stream)

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.b.record(self.stream)
            PROFILE.setdefault(self.name, []).append((self.a, self.b))


def profile_summary():
    """-> {kernel: (launches, mean_ms)} after a torch.cuda.synchronize()."""
    out = {}
    for k, pairs in (PROFILE or {}).items():
        ms = [a.elapsed_time(b) for a, b in pairs]
        out[k] = (len(ms), sum(ms) / max(1, len(ms)))
    return out


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


# ------------------------------------------------------------------------------ hash grid
def hashgrid_fwd(x: torch.Tensor, params: torch.Tensor, meta, layout=FEAT_AOS, with_jac: bool = False):
    """with_jac: also return d out / d x ([L, 3, M, 2]) for `hashgrid_dx_from_jac` (use when x needs a gradient)."""
    M = x.shape[0]
    nf = meta.n_levels * meta.n_features
    out = torch.empty((M, nf) if layout == FEAT_AOS else (meta.n_levels, M, meta.n_features),
                      dtype=torch.float32, device=x.device)
    if with_jac:
        jac = torch.empty((meta.n_levels, 3, M, 2), dtype=torch.float32, device=x.device)
        with _timed("hashgrid_fwd"):
            check(lib().mipsf_hashgrid_fwd(dptr(x), dptr(params), dptr(out), dptr(jac), M, C.byref(meta), layout,
                                           stream_ptr()), "hashgrid_fwd (with the Jacobian)")
        return out, jac
    with _timed("hashgrid_fwd"):
        check(lib().mipsf_hashgrid_fwd(dptr(x), dptr(params), dptr(out), None, M, C.byref(meta), layout, stream_ptr()),
              "hashgrid_fwd")
    return out


def hashgrid_dx_from_jac(jac, dout, dx, meta, layout=FEAT_AOS, tiles=None):
    """dx += J . dout with the Jacobian saved by hashgrid_fwd(with_jac=True).  tiles: the live-tile lists that
    ``decoder_bwd(..., return_tiles=True)`` hands back together with THIS `dout` -- the samples of the other tiles have a
    zero gradient and are left out (their Jacobian is not read).  Only pass lists that describe `dout` as it is now: a
    gradient added into `dout` afterwards (a feature regulariser, say) makes them stale -- leave `tiles` out then."""
    M = dx.shape[0]
    with _timed("hashgrid_dx"):
        check(lib().mipsf_hashgrid_dx_from_jac(dptr(jac), dptr(dout), dptr(dx), dptr(tiles, torch.int32), M,
                                               C.byref(meta), layout, stream_ptr()), "hashgrid_dx_from_jac")


_SIDE_STREAMS = {}
# decoder_bwd: short-cut the 32-sample tiles whose incoming gradient is zero throughout (exact; MIPSF_NO_TILE_SKIP=1 keeps
# every tile, for A/B measurements)
SKIP_ZERO_TILES = os.environ.get("MIPSF_NO_TILE_SKIP", "0") != "1"
_LAST_TILE_LIVE = None          # (buffer, M) of the most recent decoder_bwd that used the short cut (reporting only)


def last_live_tile_share() -> Optional[float]:
    """Share of the 32-sample tiles the most recent ``decoder_bwd`` found to carry a gradient (None: no such call yet).
    Reads the counts of the chain kernel's live-tile lists (words 64 q + 32, csrc/decoder16.hip); synchronises."""
    if _LAST_TILE_LIVE is None:
        return None
    buf, M = _LAST_TILE_LIVE
    counts = buf[32:512:64].cpu()
    return float(counts.sum().item()) / max(1, (M + 31) // 32)


_LAST_SCATTER_DOUT = None       # the feature gradient of the most recent profiled hashgrid_bwd (reporting only)


def last_live_record_share() -> Optional[float]:
    """Share of the (sample, level) pairs of the most recent PROFILED ``hashgrid_bwd`` whose two feature gradients are
    not both zero -- the pairs the routed scatter makes records for (None: no such call).  Synchronises."""
    if _LAST_SCATTER_DOUT is None:
        return None
    d = _LAST_SCATTER_DOUT.reshape(-1, 2)
    return float(((d[:, 0] != 0) | (d[:, 1] != 0)).float().mean().item())


def side_stream(device) -> "torch.cuda.Stream":
    """One helper stream per device for work that only depends on the inputs of a pass (hashgrid_route_ahead)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
    return _SIDE_STREAMS[key]


def hashgrid_route_ahead(x, meta):
    """Routing half of hashgrid_bwd (a third of its time; depends on x only) enqueued on the device's side stream, i.e.
    next to whatever the current stream does from here on.  -> (scratch, event): hand both to ``hashgrid_bwd(routed=...)``;
    the CALLER must make the current stream wait for `event` (or the side stream) before the scratch buffer's memory can
    be reused -- ``hashgrid_bwd`` does.  Works under hipGraph capture (a fork / join of the captured stream)."""
    M = x.shape[0]
    n = _lib.buffer_size(_lib.SIZE_HASHGRID_BWD_SCRATCH, M, 0, 0, meta)
    scratch = torch.empty(n, dtype=torch.float32, device=x.device)       # owned by the CURRENT stream's allocator pool
    main, side = torch.cuda.current_stream(x.device), side_stream(x.device)
    side.wait_stream(main)
    with _timed("hashgrid_route", side):
        check(lib().mipsf_hashgrid_route(dptr(x), dptr(scratch), M, C.byref(meta), C.c_void_p(side.cuda_stream)),
              "hashgrid_route")
    ev = torch.cuda.Event()
    ev.record(side)
    return scratch, ev


_ZEROED = {}
_lib.KEPT_BLOCK_CACHES.append(_ZEROED)


def _zeroed_words(device, n, tag):
    """A small int32 buffer that is zero when a kernel starts and is left at zero by it (tickets, counter blocks): one per
    (device, stream, use, size), created zero once -- calls on one stream are ordered."""
    key = (device.index, stream_ptr(), tag, n)
    buf = _ZEROED.get(key)
    if buf is None:
        buf = _ZEROED[key] = torch.zeros(n, dtype=torch.int32, device=device)
    return buf


_SCATTER_COUNTERS = {}
_lib.KEPT_BLOCK_CACHES.append(_SCATTER_COUNTERS)


def _scatter_counters(device, meta):
    """The routed scatter's counter block (bin counts, queue head, tickets): zero once, left ready by every call
    (mipsf_hashgrid_bwd), so one zero-initialised block per (device, stream, size) is kept -- calls on one stream are
    ordered -- and no call launches a clearing kernel.  One stream per block: a graph captured on stream A must not be
    replayed concurrently on two streams (both replays would count in the same block); a failed call empties the cache
    (_lib.check)."""
    n = _lib.buffer_size(_lib.SIZE_HASHGRID_COUNTER_WORDS, meta=meta)
    key = (device.index, stream_ptr(), n)
    buf = _SCATTER_COUNTERS.get(key)
    if buf is None:
        buf = _SCATTER_COUNTERS[key] = torch.zeros(n, dtype=torch.int32, device=device)
    return buf


# the chain kernel leaves out the half of its gradient record that the weight-gradient kernel can recompute (experiments: 0)
LEAN_DACT = not bool(os.environ.get("MIPSF_FULL_DACT"))
HG_DPARAMS_ZERO = 1     # include/mipsf.h MIPSF_HG_DPARAMS_ZERO
HG_ROUTED = 2           # MIPSF_HG_ROUTED
_HG_IGNORE_ZERO_HINT = bool(os.environ.get("MIPSF_HG_IGNORE_ZERO_HINT"))     # experiments: always read-modify-write


def hashgrid_bwd(x, params, dout, dparams, meta, layout=FEAT_AOS, dx: Optional[torch.Tensor] = None, routed=None,
                 dparams_zero=False):
    """dparams (and dx when given) are accumulated into.  routed: (scratch, event) of hashgrid_route_ahead for this x.
    dparams_zero: the caller vouches that dparams is all zero now (a fresh torch.zeros, a gradient buffer the optimiser
    cleared): the table slices are stored instead of read-modify-written."""
    M = x.shape[0]
    if PROFILE is not None:
        global _LAST_SCATTER_DOUT
        _LAST_SCATTER_DOUT = dout
    a = _lib.HashgridBwdArgs.new(M=M, x=dptr(x), params=dptr(params), dout=dptr(dout), dparams=dptr(dparams),
                                 meta=C.pointer(meta), feat_layout=layout)
    if routed is not None and dx is None and dparams is not None:
        scratch, ev = routed
        torch.cuda.current_stream(x.device).wait_event(ev)
        a.scratch, a.flags = dptr(scratch), HG_ROUTED
        with _timed("hashgrid_bwd"):
            check(lib().mipsf_hashgrid_bwd(C.byref(a), stream_ptr()), "hashgrid_bwd (routed)")
        return
    n = _lib.buffer_size(_lib.SIZE_HASHGRID_BWD_SCRATCH, M, 1 if dx is not None else 0, 0, meta)
    scratch = torch.empty(n, dtype=torch.float32, device=x.device)
    counters = _scatter_counters(x.device, meta)
    a.dx, a.scratch, a.counters = dptr(dx), dptr(scratch), dptr(counters, torch.int32)
    a.flags = HG_DPARAMS_ZERO if (dparams_zero and not _HG_IGNORE_ZERO_HINT) else 0
    with _timed("hashgrid_bwd"):
        check(lib().mipsf_hashgrid_bwd(C.byref(a), stream_ptr()), "hashgrid_bwd")


def hashgrid_indices(x, meta) -> torch.Tensor:
    M = x.shape[0]
    idx = torch.empty((M, meta.n_levels, 8), dtype=torch.int32, device=x.device)
    check(lib().mipsf_hashgrid_indices(dptr(x), dptr(idx, torch.int32), M, C.byref(meta), stream_ptr()),
          "hashgrid_indices")
    return idx


class HashGridFn(torch.autograd.Function):
    """tcnn.Encoding(HashGrid) forward/backward (dense fp32 parameter gradient, dL/dx)."""

    @staticmethod
    def forward(ctx, x, params, meta):
        x = _f32c(x)
        ctx.meta = meta
        if ctx.needs_input_grad[0]:
            out, jac = hashgrid_fwd(x, params.detach(), meta, FEAT_AOS, with_jac=True)
            ctx.save_for_backward(x, params, jac)
            return out
        ctx.save_for_backward(x, params)
        return hashgrid_fwd(x, params.detach(), meta, FEAT_AOS)

    @staticmethod
    def backward(ctx, dout):
        x, params, *jac = ctx.saved_tensors    # saved tensors survive retain_graph=True re-entry
        dout = _f32c(dout)
        dparams = torch.zeros_like(params) if ctx.needs_input_grad[1] else None
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        if dparams is not None:
            hashgrid_bwd(x, params.detach(), dout, dparams, ctx.meta, FEAT_AOS, None, dparams_zero=True)
        if dx is not None:
            hashgrid_dx_from_jac(jac[0], dout, dx, ctx.meta, FEAT_AOS)
        return dx, dparams, None


# ------------------------------------------------------------------------------ frequency
class FrequencyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_freq):
        x = _f32c(x)
        ctx.n_freq = n_freq
        ctx.save_for_backward(x)
        M, D = x.shape
        out = torch.empty((M, D * 2 * n_freq), dtype=torch.float32, device=x.device)
        check(lib().mipsf_freq_fwd(dptr(x), dptr(out), M, D, n_freq, stream_ptr()), "freq_fwd")
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        M, D = x.shape
        dx = torch.empty_like(x)
        check(lib().mipsf_freq_bwd(dptr(x), dptr(_f32c(dout)), dptr(dx), M, D, ctx.n_freq, stream_ptr()),
              "freq_bwd")
        return dx, None


# -------------------------------------------------------------------------------- decoder
def _decoder_struct(tensors, cls):
    st = cls()
    for field, t, shape in zip(_DEC_FIELDS, tensors, _DEC_SHAPES):
        if tuple(t.shape) != shape:
            raise RuntimeError(f"decoder tensor {field} has shape {tuple(t.shape)}, expected {shape}")
        setattr(st, field, dptr(t))
    return st


def decoder_pack(weights, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """weights: the 10 nn.Linear tensors in DECODER_PARAM_ORDER -> MFMA operand images."""
    dev = weights[0].device
    if packed is None:
        packed = torch.empty(_lib.buffer_size(_lib.SIZE_DECODER_PACKED), dtype=torch.float32, device=dev)
    ws = [w.detach() for w in weights]
    st = _decoder_struct(ws, _lib.DecoderWeights)
    with _timed("decoder_pack"):
        check(lib().mipsf_decoder_pack(C.byref(st), dptr(packed), stream_ptr()), "decoder_pack")
    return packed


def decoder_pack16(weights, packed16: Optional[torch.Tensor] = None, precision: str = "f16x3") -> torch.Tensor:
    """weights: the 10 nn.Linear tensors in DECODER_PARAM_ORDER -> 16-bit operand images of the three hidden layers and
    the two heads, forward and backward sets.  precision "f16x3" / "f16": hi + lo f16 halves (one buffer serves both);
    "bf16x6": the three bf16 pieces of every weight (another, larger buffer: it is for the bf16x6 kernels only)."""
    dev = weights[0].device
    fam = _PACK16_FAMILY[precision]
    n = _lib.buffer_size(_lib.SIZE_DECODER_PACKED16, 0, _lib.PREC[fam])
    if packed16 is None:
        packed16 = torch.empty(n, dtype=torch.float32, device=dev)
    elif packed16.numel() != n:
        raise RuntimeError(f"packed16 has {packed16.numel()} floats, precision {precision!r} needs {n}")
    st = _decoder_struct([w.detach() for w in weights], _lib.DecoderWeights)
    with _timed("decoder_pack"):
        check(lib().mipsf_decoder_pack16(C.byref(st), dptr(packed16), _lib.PREC[fam], stream_ptr()), "decoder_pack16")
    packed16.mipsf_family = fam
    return packed16


_PACK16_FAMILY = {"f16x3": "f16x3", "f16": "f16x3", "bf16x6": "bf16x6"}
SPLIT_PRECISIONS = ("f16x3", "bf16x6")        # decoder arithmetics that keep activations (training)


def _check_family(packed16, precision):
    """An operand buffer packed for one family must not reach the other's kernels (they would read an f16 buffer past its end,
    or take bf16 planes for f16 halves).  The buffers differ in SIZE, which survives views, clones and .to() -- the python
    attribute is only the better message; the library checks the size again (packed16_floats of its argument blocks)."""
    fam = getattr(packed16, "mipsf_family", None)
    if fam is not None and fam != _PACK16_FAMILY[precision]:
        raise RuntimeError(f"packed16 was packed for {fam!r}, the call asks for {precision!r}")
    n = _lib.buffer_size(_lib.SIZE_DECODER_PACKED16, 0, _lib.PREC[_PACK16_FAMILY[precision]])
    if packed16.numel() != n:
        raise RuntimeError(f"packed16 holds {packed16.numel()} floats, precision {precision!r} needs {n}: packed for the other family?")


def decoder_fwd(packed, feat, layout, x, embed_pos, M, save, precision: str = "f32", packed16=None):
    """precision "f32": fp32-input MFMA (exact fp32 products).  "f16x3": f16 MFMA on hi/lo split operands (~3e-7
    relative; same `saved` layout, so the backward kernels are unchanged).  "f16": plain f16 operands, forward only.
    The f16 modes need `packed16` (decoder_pack16) and the in-kernel positional encoding (embed_pos None).
    save: False, True (the full activation record), "lean" (f16x3: H1, a third of the record, is not written -- only
    valid with ``decoder_bwd(..., wgrad_precision="stream_f16x3", recompute_h1=True)``, which recomputes it from x) or
    "masks" (f16x3: only the ReLU masks, 32 B per sample -- all the backward CHAIN reads; ``decoder_bwd(grads=None)`` only)."""
    lean = save in ("lean", "masks")
    masks_only = save == "masks"
    if lean and precision not in SPLIT_PRECISIONS:
        raise RuntimeError('save="lean" / "masks" belong to precision "f16x3" / "bf16x6"')
    out = torch.empty((M, 10), dtype=torch.float32, device=x.device)
    saved = None
    if save:
        saved = torch.empty(_lib.buffer_size(_lib.SIZE_DECODER_SAVED, M), dtype=torch.float32, device=x.device)
    if precision != "f32":
        if embed_pos is not None or packed16 is None:
            raise RuntimeError("the f16 decoder modes take packed16 and compute the positional encoding in-kernel")
        if save and precision not in SPLIT_PRECISIONS:
            raise RuntimeError('only precision "f32" / "f16x3" / "bf16x6" keep activations for the backward pass')
        _check_family(packed16, precision)
        # the live-tile lists of the backward chain that will follow this record: allocated here so that THIS launch clears
        # their counters (a memset in front of the chain kernel was a launch of its own)
        tile_live = None
        if save and SKIP_ZERO_TILES:
            tile_live = torch.empty(_lib.buffer_size(_lib.SIZE_DECODER_TILE_WORDS, M), dtype=torch.int32, device=x.device)
        a = _lib.DecoderFwd16Args.new(M=M, packed16=dptr(packed16), feat=dptr(feat), x=dptr(x), out=dptr(out), saved=dptr(saved),
                                      tile_live_clear=dptr(tile_live, torch.int32), feat_layout=layout,
                                      precision=_lib.PREC[precision], lean_record=(2 if masks_only else 1) if lean else 0,
                                      packed16_floats=packed16.numel())
        with _timed("decoder_fwd"):
            check(lib().mipsf_decoder_fwd16(C.byref(a), stream_ptr()), "decoder_fwd16")
        if lean:
            saved.mipsf_lean_record = True          # decoder_bwd refuses to read H1 from such a record
        if masks_only:
            saved.mipsf_masks_only = True           # ... and to compute weight gradients at all from this one
        if tile_live is not None:
            saved.mipsf_tile_live = [tile_live, True]      # (buffer, its counters are still clear)
        return out, saved
    pe_mode = 0 if embed_pos is None else 1
    with _timed("decoder_fwd"):
        check(lib().mipsf_decoder_fwd(dptr(packed), dptr(feat), layout, dptr(x), dptr(embed_pos), pe_mode, dptr(out),
                                      dptr(saved), M, stream_ptr()), "decoder_fwd")
    return out, saved


def decoder_bwd(packed, feat, layout, x, embed_pos, out, dout, saved, grads, M, precision: str = "f32", packed16=None,
                wgrad_precision: str = "auto", recompute_h1: bool = False, return_tiles: bool = False):
    """grads: 10 tensors in DECODER_PARAM_ORDER, accumulated into, or None (frozen decoder: the weight-gradient
    GEMMs are skipped).  -> (dfeat, dx, dembed_pos|None).  precision "f16x3": the activation-gradient chain runs on
    the f16 matrix cores with hi/lo split operands (packed16, in-kernel positional encoding); it leaves the same `dact`
    record as the fp32 chain, so every weight-gradient kernel works behind either.  wgrad_precision:
      "stream_f16x3"  streaming kernel (csrc/wgrad16.hip), hi + lo f16 planes, a power-of-two scale per gradient block:
                      fp32-class (~3e-7 of the maximum), runs at the device's read bandwidth; pe_mode 0
      "stream_bf16x6" the same kernel on three bf16 planes (no scale; fp32-class, slower), "stream_bf16x3" two planes (~5e-6)
      "f32"           LDS-transposing kernel on the fp32-input matrix cores (the round-1 path; any pe_mode)
      "bf16x3"        that kernel with bf16 hi/lo operands for its three large products (~5e-6)
      "auto"          "stream_f16x3" behind the f16x3 chain, "f32" otherwise.
    recompute_h1 (stream_f16x3 + packed16): H1 is recomputed from x instead of read from `saved` -- required when the
    forward kept the lean record (``decoder_fwd(save="lean")``), bit-identical otherwise.
    Zero tiles (SKIP_ZERO_TILES, f16x3 chain with the streaming weight-gradient kernel or a frozen decoder): 32-sample
    tiles whose incoming gradient is zero throughout -- the ray tails behind the truncation band, a third of a mapping
    batch -- are flagged by the chain, get zero dfeat / dx, and are never touched by the weight-gradient kernel.
    return_tiles: a 4th return value, the chain's live-tile lists (or None when no short cut was taken), for
    ``hashgrid_dx_from_jac(..., tiles=)``."""
    if wgrad_precision == "auto":
        wgrad_precision = ("stream_" + precision) if (precision in SPLIT_PRECISIONS and embed_pos is None) else "f32"
    if getattr(saved, "mipsf_masks_only", False) and grads is not None:
        raise RuntimeError("this activation record holds the ReLU masks only (decoder_fwd(save='masks')): no weight gradients")
    if getattr(saved, "mipsf_lean_record", False) and grads is not None:
        if wgrad_precision not in ("stream_f16x3", "stream_bf16x6") or packed16 is None or wgrad_precision != "stream_" + precision:
            raise RuntimeError("this activation record was saved lean (no H1): the weight gradients need the streaming kernel "
                               "of the forward's arithmetic (wgrad_precision='stream_f16x3' / 'stream_bf16x6') with packed16 "
                               "(H1 is recomputed from x)")
        recompute_h1 = True
    dev = x.device
    dfeat = torch.empty_like(feat)
    dx = torch.empty((M, 3), dtype=torch.float32, device=dev)
    dpe = torch.empty((M, 48), dtype=torch.float32, device=dev) if embed_pos is not None else None
    # (a frozen decoder behind the f16x3 chain: the pre-activation gradients are for the weight-gradient kernel only)
    dact = None if (grads is None and precision in SPLIT_PRECISIONS) else torch.empty(_lib.buffer_size(_lib.SIZE_DECODER_DACT, M), dtype=torch.float32, device=dev)
    pe_mode = 0 if embed_pos is None else 1
    tile_live = None
    lean_dact = False
    if precision in SPLIT_PRECISIONS:
        if embed_pos is not None or packed16 is None:
            raise RuntimeError("the f16x3 / bf16x6 backward chain takes packed16 and computes the positional encoding in-kernel")
        _check_family(packed16, precision)
        hdr_clear = 4 if precision == "bf16x6" else 0        # MIPSF_CHAIN_BF16X6
        # the lean gradient record (half of `dact`): when the exchange form of the streaming f16 kernel follows, which
        # recomputes dG3 and the rgb_emb half of dH2 from the small rows + the ReLU masks (csrc/wgrad16.hip)
        lean_dact = bool(LEAN_DACT and grads is not None and wgrad_precision == "stream_" + precision and recompute_h1
                         and packed16 is not None)
        if SKIP_ZERO_TILES and (grads is None or wgrad_precision.startswith("stream_")):
            global _LAST_TILE_LIVE
            pre = getattr(saved, "mipsf_tile_live", None)
            if pre is not None and pre[1]:          # the forward of this record cleared the counters: use its buffer once
                tile_live, hdr_clear = pre[0], hdr_clear | 1
                pre[1] = False                      # (a second backward through the same record clears them itself)
            else:
                tile_live = torch.empty(_lib.buffer_size(_lib.SIZE_DECODER_TILE_WORDS, M), dtype=torch.int32, device=dev)
            _LAST_TILE_LIVE = (tile_live, M)
        a = _lib.DecoderChain16Args.new(M=M, packed16=dptr(packed16), x=dptr(x), out=dptr(out), dout=dptr(dout), saved=dptr(saved),
                                        dfeat=dptr(dfeat), dx=dptr(dx), dact=dptr(dact), tile_live=dptr(tile_live, torch.int32),
                                        feat_layout=layout, flags=hdr_clear | (2 if lean_dact else 0), packed16_floats=packed16.numel())
        with _timed("decoder_bwd_chain"):
            check(lib().mipsf_decoder_bwd_chain16(C.byref(a), stream_ptr()), "decoder_bwd_chain16")
    else:
        with _timed("decoder_bwd_chain"):
            check(lib().mipsf_decoder_bwd_chain(dptr(packed), layout, dptr(x), pe_mode, dptr(out), dptr(dout),
                                                dptr(saved), dptr(dfeat), dptr(dx), dptr(dpe), dptr(dact), M,
                                                stream_ptr()), "decoder_bwd_chain")
    if grads is not None:
        partial = torch.empty(_lib.buffer_size(_lib.SIZE_DECODER_WGRAD_PARTIAL), dtype=torch.float32, device=dev)
        st = _decoder_struct(grads, _lib.DecoderGrads)
        if wgrad_precision.startswith("stream_"):
            # streaming kernel (csrc/wgrad16.hip): "stream_f16x3" (default of the f16x3 decoder), "stream_bf16x6", "stream_bf16x3"
            if embed_pos is not None:
                raise RuntimeError("the streaming weight-gradient kernel computes the positional encoding in-kernel")
            arith = _lib.PREC[wgrad_precision[len("stream_"):]]
            if recompute_h1 and (wgrad_precision not in ("stream_f16x3", "stream_bf16x6") or packed16 is None):
                raise RuntimeError("recompute_h1 needs wgrad_precision 'stream_f16x3' / 'stream_bf16x6' and packed16")
            if recompute_h1:
                _check_family(packed16, wgrad_precision[len("stream_"):])
            a = _lib.DecoderWgrad16Args.new(M=M, packed16=dptr(packed16) if recompute_h1 else None, feat=dptr(feat), x=dptr(x),
                                            saved=dptr(saved), dact=dptr(dact), tile_live=dptr(tile_live, torch.int32),
                                            grads=C.pointer(st), partial=dptr(partial), feat_layout=layout, arithmetic=arith,
                                            flags=1 if lean_dact else 0, packed16_floats=packed16.numel() if recompute_h1 else 0)
            with _timed("decoder_wgrad"):
                check(lib().mipsf_decoder_wgrad16(C.byref(a), stream_ptr()), "decoder_wgrad16")
            return (dfeat, dx, dpe, tile_live) if return_tiles else (dfeat, dx, dpe)
        if recompute_h1:
            raise RuntimeError("recompute_h1 needs wgrad_precision 'stream_f16x3'")
        with _timed("decoder_wgrad"):
            wprec = _lib.PREC[wgrad_precision]
            check(lib().mipsf_decoder_wgrad(dptr(feat), layout, dptr(x), dptr(embed_pos), pe_mode, dptr(saved),
                                            dptr(dact), C.byref(st), dptr(partial), wprec, M, stream_ptr()),
                  "decoder_wgrad")
    return (dfeat, dx, dpe, tile_live) if return_tiles else (dfeat, dx, dpe)


def decoder_fwd_sdf(packed, feat, layout, x, embed_pos, M, precision: str = "f32", packed16=None) -> torch.Tensor:
    """SDF column only (JointEncoding.query_sdf, model/scene_rep.py:106-107): [M] floats, bit-identical to column 3 of
    decoder_fwd, without the rgb half of layer 2, the rgb head and nine tenths of the output."""
    sdf = torch.empty((M,), dtype=torch.float32, device=x.device)
    if precision != "f32":
        if embed_pos is not None or packed16 is None:
            raise RuntimeError("the f16 decoder modes take packed16 and compute the positional encoding in-kernel")
        _check_family(packed16, precision)
        with _timed("decoder_fwd"):
            a = _lib.DecoderFwd16Args.new(M=M, packed16=dptr(packed16), feat=dptr(feat), x=dptr(x), out=dptr(sdf), feat_layout=layout,
                                          precision=_lib.PREC[precision], sdf_only=1, packed16_floats=packed16.numel())
            check(lib().mipsf_decoder_fwd16(C.byref(a), stream_ptr()), "decoder_fwd16")
        return sdf
    pe_mode = 0 if embed_pos is None else 1
    with _timed("decoder_fwd"):
        check(lib().mipsf_decoder_fwd_sdf(dptr(packed), dptr(feat), layout, dptr(x), dptr(embed_pos), pe_mode, dptr(sdf),
                                          M, stream_ptr()), "decoder_fwd_sdf")
    return sdf


class DecoderFn(torch.autograd.Function):
    """MLP_reg.forward(embed, embed_pos, query_pts) with all three inputs differentiable (module API)."""

    @staticmethod
    def forward(ctx, embed, embed_pos, x, *weights):
        embed, embed_pos, x = _f32c(embed), _f32c(embed_pos), _f32c(x)
        M = x.shape[0]
        packed = decoder_pack(weights)
        need = any(ctx.needs_input_grad)
        out, saved = decoder_fwd(packed, embed, FEAT_AOS, x, embed_pos, M, save=need)
        ctx.M = M
        ctx.save_for_backward(embed, embed_pos, x, out, saved, packed, *weights)
        return out

    @staticmethod
    def backward(ctx, dout):
        embed, embed_pos, x, out, saved, packed, *weights = ctx.saved_tensors
        need_w = any(ctx.needs_input_grad[3:])
        grads = [torch.zeros_like(w) for w in weights] if need_w else None
        dfeat, dx, dpe = decoder_bwd(packed, embed, FEAT_AOS, x, embed_pos, out, _f32c(dout), saved, grads, ctx.M)
        return (dfeat, dpe, dx, *(grads if need_w else [None] * len(weights)))


# ------------------------------------------------------------------------------- renderer
def make_render_cfg(cfg: dict, bound64, half_len64, n_uniform: int, n_near: int, emd_w: float) -> _lib.RenderCfg:
    tr, cam = cfg["training"], cfg["cam"]
    rc = _lib.RenderCfg()
    rc.n_uniform, rc.n_near = n_uniform, n_near
    rc.perturb = 1 if tr["perturb"] > 0.0 else 0
    rc.use_bound = 1 if cfg["grid"]["use_bound_normalize"] else 0
    for d in range(3):
        rc.bound_min[d] = float(bound64[d][0])
        rc.bound_max[d] = float(bound64[d][1])
        rc.half_len[d] = float(half_len64[d])
    if not cfg["grid"]["tcnn_encoding"]:
        raise RuntimeError("grid.tcnn_encoding = False is not a configuration the reference ships")
    rc.norm_factor = float(tr["norm_factor"])
    rc.trunc = float(tr["trunc"])
    rc.sc_factor = float(cfg["data"]["sc_factor"])
    rc.depth_trunc = float(cam["depth_trunc"])
    rc.rgb_missing_nonzero = 1 if tr["rgb_missing"] != 0 else 0
    rc.emd_w = float(emd_w)
    return rc


def normalise_points(pts: torch.Tensor, rc) -> torch.Tensor:
    xn = torch.empty_like(pts)
    check(lib().mipsf_normalise_points(dptr(pts), C.byref(rc), dptr(xn), pts.shape[0], stream_ptr()), "normalise")
    return xn


def normalise_bwd(dxn: torch.Tensor, rc) -> torch.Tensor:
    dpts = torch.empty_like(dxn)
    check(lib().mipsf_normalise_bwd(dptr(dxn), C.byref(rc), dptr(dpts), dxn.shape[0], stream_ptr()), "normalise_bwd")
    return dpts


def sample_rays(rays_o, rays_d, target_d, noise, tables, rc, N, S):
    dev = rays_o.device
    z_vals = torch.empty((N, S), dtype=torch.float32, device=dev)
    xn = torch.empty((N * S, 3), dtype=torch.float32, device=dev)
    counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
    zu, zoff, znd = tables
    with _timed("sample_rays"):
        check(lib().mipsf_sample_rays(dptr(rays_o), dptr(rays_d), dptr(target_d), dptr(noise), dptr(zu), dptr(zoff),
                                      dptr(znd), C.byref(rc), dptr(z_vals), dptr(xn), dptr(counts, torch.int32), N,
                                      stream_ptr()), "sample_rays")
    return z_vals, xn, counts


def _render_fwd_args(raw, z_vals, target_rgb, target_d, counts, rc, rgb, depth, var, disp, acc, weights, N, S):
    return _lib.RenderFwdArgs.new(N=N, S=S, raw=dptr(raw), z_vals=dptr(z_vals), target_rgb=dptr(target_rgb), target_d=dptr(target_d),
                                  counts=dptr(counts, torch.int32) if counts is not None else None, cfg=C.pointer(rc),
                                  rgb=dptr(rgb), depth=dptr(depth), depth_var=dptr(var), disp=dptr(disp), acc=dptr(acc),
                                  weights=dptr(weights))


RENDER_DRAW_MAX_S = 128     # mipsf_render_fwd's `draw`: two samples per lane

_UNIT_GRAD = {}


def unit_grad(device):
    """THE root gradient of a plain ``loss.backward()``: one cached fp32 scalar 1.0 per device (helper_functions.utils.
    backward_from_one passes it).  ``_RenderFn.backward`` recognises it by address -- the gradient of the objective is then known
    to be exactly 1 on the host, and the gradient the forward launch has already written is returned without a launch.  READ-ONLY:
    a caller that scales it in place (loss scaling) turns the shortcut off for good -- the tensor's version counter is checked,
    and the device-side check of ``MIPSF_RENDER_BWD_KEEP_IF_UNIT`` decides from then on."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    one = _UNIT_GRAD.get(dev)
    if one is None:
        one = _UNIT_GRAD[dev] = torch.ones((), dtype=torch.float32, device=dev)
    return one


def render_fwd(raw, z_vals, target_rgb, target_d, counts, rc, N, S, train: bool, want_weights=False, loss_weights=None,
               share_of=None, want_draw=False):
    """loss_weights (train only): device tensor of the 4 loss weights -> an 8th return value, the objective
    sum_k w_k * losses[k] formed inside the loss kernel (one float).
    share_of (train only): ``reduce(t)`` -- these N rays are ONE SHARE of a ray-data-parallel batch; ``reduce`` receives the
    share's ten fp64 numbers (the nine loss sums + its ray count) and must return their sums over all shares (an all-reduce
    of 80 bytes); the losses are those of the WHOLE batch, identical on every rank, and a 9th value is returned: the whole
    batch's ray count (for render_bwd's n_norm).
    want_draw (train with loss_weights, no share_of, S <= RENDER_DRAW_MAX_S): the same launch also writes d objective / d raw for
    an objective gradient of exactly 1 -- one more return value, [N, S, 10] (render_bwd(keep_draw=) decides whether it stands)."""
    dev = raw.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)   # noqa: E731
    rgb, depth, var, disp, acc = f(N, 3), f(N), f(N), f(N), f(N)
    weights = f(N, S) if want_weights else None
    losses = f(8) if train else None
    partial = f(_lib.buffer_size(_lib.SIZE_RENDER_PARTIAL, N)) if train else None
    total = f(1) if (train and loss_weights is not None) else None
    ticket = _zeroed_words(dev, 1, "render_fwd") if train else None    # the last workgroup finishes the losses: one launch
    if train and share_of is not None:
        sums = torch.zeros(10, dtype=torch.float64, device=dev)
        a = _render_fwd_args(raw, z_vals, target_rgb, target_d, counts, rc, rgb, depth, var, disp, acc, weights, N, S)
        a.partial, a.sums, a.ticket = dptr(partial), dptr(sums, torch.float64), dptr(ticket, torch.int32)
        with _timed("render_fwd"):
            check(lib().mipsf_render_fwd(C.byref(a), stream_ptr()), "render_fwd (sums of a share)")
        sums[9] = float(N)
        sums = share_of(sums)
        n_total = int(round(float(sums[9])))            # (one small read-back per step: the collective synchronises anyway)
        check(lib().mipsf_loss_finalize_sums(dptr(sums, torch.float64), C.byref(rc), n_total, S, dptr(losses),
                                             dptr(loss_weights) if total is not None else None, dptr(total), stream_ptr()),
              "loss_finalize_sums")
        if loss_weights is not None:
            return rgb, depth, var, disp, acc, weights, losses, total, n_total
        return rgb, depth, var, disp, acc, weights, losses, None, n_total
    a = _render_fwd_args(raw, z_vals, target_rgb, target_d, counts, rc, rgb, depth, var, disp, acc, weights, N, S)
    a.losses, a.partial, a.ticket = dptr(losses), dptr(partial), dptr(ticket, torch.int32)
    a.loss_weights, a.loss_total = (dptr(loss_weights) if total is not None else None), dptr(total)
    draw = None
    if want_draw:
        if total is None or S > RENDER_DRAW_MAX_S:
            raise ValueError(f"want_draw needs the training objective (loss_weights) and S <= {RENDER_DRAW_MAX_S}")
        draw = torch.empty_like(raw)
        a.draw = dptr(draw)
    with _timed("render_fwd"):
        check(lib().mipsf_render_fwd(C.byref(a), stream_ptr()), "render_fwd")
    if want_draw:
        return rgb, depth, var, disp, acc, weights, losses, total, draw
    if loss_weights is not None and train:
        return rgb, depth, var, disp, acc, weights, losses, total
    return rgb, depth, var, disp, acc, weights, losses


def render_bwd(raw, z_vals, target_rgb, target_d, counts, losses, rc, g_losses, g_rgb, g_depth, N, S, g_total=None,
               loss_weights=None, n_norm=None, keep_draw=None):
    """g_total / loss_weights: gradient of render_fwd's objective and its weights (the kernel forms g_total * w itself).
    n_norm: the ray count the losses were normalised by when these N rays are a share of a larger batch (render_fwd(share_of=)).
    keep_draw: render_fwd(want_draw=True)'s gradient buffer -- it is returned; the kernel leaves it alone when g_total is exactly
    1 (decided on the device) and rewrites it otherwise.  g_total must then be the only gradient."""
    draw = torch.empty_like(raw) if keep_draw is None else keep_draw
    a = _lib.RenderBwdArgs.new(N=N, S=S, N_norm=0 if n_norm is None else int(n_norm), raw=dptr(raw), z_vals=dptr(z_vals),
                               target_rgb=dptr(target_rgb), target_d=dptr(target_d),
                               counts=dptr(counts, torch.int32) if counts is not None else None, losses=dptr(losses),
                               cfg=C.pointer(rc), g_losses=dptr(g_losses), g_total=dptr(g_total),
                               loss_weights=dptr(loss_weights) if g_total is not None else None, g_rgb=dptr(g_rgb),
                               g_depth=dptr(g_depth), draw=dptr(draw),
                               flags=_lib.RENDER_BWD_KEEP_IF_UNIT if keep_draw is not None else 0)
    with _timed("render_bwd"):
        check(lib().mipsf_render_bwd(C.byref(a), stream_ptr()), "render_bwd")
    return draw


def rays_bwd(dxn, z_vals, rc, N, S):
    d_o = torch.empty((N, 3), dtype=torch.float32, device=dxn.device)
    d_d = torch.empty((N, 3), dtype=torch.float32, device=dxn.device)
    with _timed("rays_bwd"):
        check(lib().mipsf_rays_bwd(dptr(dxn), dptr(z_vals), C.byref(rc), dptr(d_o), dptr(d_d), N, S, stream_ptr()),
              "rays_bwd")
    return d_o, d_d


# ------------------------------------------------------------------------- rays from poses
_POSE_SCRATCH = {}
_lib.KEPT_BLOCK_CACHES.append(_POSE_SCRATCH)


def _pose_scratch(device, F, K, N):
    """Scratch of mipsf_pose_rays_bwd: its first word is a ticket that must be zero on entry and is left zero by the
    kernel, so one zero-initialised buffer per (device, stream, size) is kept and reused (calls on one stream are
    ordered; no per-call memset launch)."""
    n = _lib.buffer_size(_lib.SIZE_POSE_RAYS_SCRATCH, N, F, K)
    key = (device.index, stream_ptr(), n)
    buf = _POSE_SCRATCH.get(key)
    if buf is None:
        buf = _POSE_SCRATCH[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return buf


def pose_handover(src, rot_out, trans_out, quaternion: bool = False):
    """One pose from stage to stage on the device.  src: 12 floats [3x3 row-major | t] (the RandomOptimizer's state), or with
    quaternion=True 7 floats [w x y z | t] -> rot_out[4] (unit quaternion, w >= 0), trans_out[3] (views of the next stage's
    pose Parameters): qt_to_transform_matrix / matrix_to_quaternion as the reference applies them between its stages, in IEEE
    fp32 -- bit-identical to the frame loop's host helpers; no host round trip."""
    check(lib().mipsf_pose_handover(dptr(src), 1 if quaternion else 0, dptr(rot_out), dptr(trans_out), stream_ptr()), "pose_handover")


class PoseRaysFn(torch.autograd.Function):
    """(rot [K,4], trans [K,3]) -> rays_o, rays_d [N,3]; fused replacement of
    ``qt_to_transform_matrix`` + ``poses_all[owner]`` gather + ``sum(d_cam * R, -1)`` (mipsfusion.py:320-322)."""

    @staticmethod
    def forward(ctx, rot, trans, fixed, owner, d_cam, in_place=False, table=None, rows=None):
        """table / rows given: d_cam is None and the rows of the ray table are gathered in the same launch
        (-> rays_o, rays_d, rgb, depth [N,1])."""
        ctx.params = (rot, trans) if in_place else None
        rot, trans = _f32c(rot), _f32c(trans)
        fixed = _f32c(fixed) if fixed is not None and fixed.numel() else None
        owner = owner.to(torch.int64).contiguous()
        F = 0 if fixed is None else fixed.shape[0]
        K = rot.shape[0]
        dev = rot.device
        if table is not None:
            flat = table.reshape(-1, 7)
            N = rows.shape[0]
            d_cam = torch.empty((N, 3), dtype=torch.float32, device=dev)
            rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
            depth = torch.empty((N, 1), dtype=torch.float32, device=dev)
        else:
            d_cam = _f32c(d_cam)
            N = d_cam.shape[0]
        rays_o = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rays_d = torch.empty((N, 3), dtype=torch.float32, device=dev)
        with _timed("pose_rays_fwd"):
            if table is not None:
                check(lib().mipsf_pose_rays_fwd(dptr(flat), flat.shape[0], dptr(rows, torch.int64), dptr(fixed),
                                                dptr(rot), dptr(trans), F, K, dptr(owner, torch.int64), dptr(d_cam),
                                                dptr(rgb), dptr(depth), dptr(rays_o), dptr(rays_d), N, stream_ptr()),
                      "pose_rays_fwd (rows of the ray table)")
            else:
                check(lib().mipsf_pose_rays_fwd(None, 0, None, dptr(fixed), dptr(rot), dptr(trans), F, K,
                                                dptr(owner, torch.int64), dptr(d_cam), None, None, dptr(rays_o), dptr(rays_d), N,
                                                stream_ptr()), "pose_rays_fwd")
        ctx.F, ctx.K, ctx.N = F, K, N
        ctx.save_for_backward(rot, owner, d_cam)
        ctx.set_materialize_grads(False)
        if table is not None:
            ctx.mark_non_differentiable(rgb, depth)
            return rays_o, rays_d, rgb, depth
        return rays_o, rays_d

    @staticmethod
    def backward(ctx, g_o, g_d, _g_rgb=None, _g_depth=None):
        rot, owner, d_cam = ctx.saved_tensors
        scratch = _pose_scratch(rot.device, ctx.F, ctx.K, ctx.N)
        g_o = _f32c(g_o) if g_o is not None else None
        g_d = _f32c(g_d) if g_d is not None else None
        # opt-in (`pose_rays(..., accumulate_in_place=True)`, plain loss.backward() loops): the kernel adds straight into
        # the leaf parameters' .grad -- no two `grad += new` passes of autograd's AccumulateGrad per iteration
        direct = ctx.params is not None and all(
            p.is_leaf and p.requires_grad and not p._backward_hooks and p.dtype == torch.float32 and p.is_contiguous()
            for p in ctx.params)
        if direct:
            for p in ctx.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            d_rot, d_trans = ctx.params[0].grad, ctx.params[1].grad
        else:
            d_rot = torch.empty((ctx.K, 4), dtype=torch.float32, device=rot.device)
            d_trans = torch.empty((ctx.K, 3), dtype=torch.float32, device=rot.device)
        with _timed("pose_rays_bwd"):
            check(lib().mipsf_pose_rays_bwd(dptr(g_o), dptr(g_d), dptr(rot), ctx.F, ctx.K, dptr(owner, torch.int64),
                                            dptr(d_cam), dptr(d_rot), dptr(d_trans), dptr(scratch), ctx.N,
                                            1 if direct else 0, stream_ptr()), "pose_rays_bwd")
        if direct:
            return None, None, None, None, None, None, None, None
        return d_rot, d_trans, None, None, None, None, None, None


def pose_rays(rot, trans, fixed_poses, owner, d_cam, accumulate_in_place=False):
    """rays_o, rays_d for rays whose camera pose is poses_all[owner] with
    poses_all = cat([fixed_poses, qt_to_transform_matrix(rot, trans)]).  accumulate_in_place (opt-in, only valid for plain
    ``loss.backward()`` accumulation loops, ignored for non-leaf / hooked parameters): the backward adds the pose
    gradients straight into ``rot.grad`` / ``trans.grad``."""
    return PoseRaysFn.apply(rot, trans, fixed_poses, owner, d_cam, accumulate_in_place)


def gather_pose_rays(table, rows, rot, trans, fixed_poses, owner, accumulate_in_place=False):
    """``gather_rays(table, rows, split=True)`` + ``pose_rays(...)`` in one launch (each is a 5 us launch of every
    iteration): rows int64 [N] (device) of the ray table [..., 7] -> rays_o, rays_d, target rgb [N,3], target depth [N,1]."""
    return PoseRaysFn.apply(rot, trans, fixed_poses, owner, None, accumulate_in_place, table, rows.to(torch.int64).contiguous())


class GatherPosePlaceFn(torch.autograd.Function):
    """Row gather of the ray table + rays from the pose Parameters + sample placement (``gather_pose_rays`` followed by
    ``JointEncoding``'s placement: keyframeSet.py:264-290, mipsfusion.py:320-322, scene_rep.py:156-179) as ONE launch
    each way: -> (rgb [N,3], depth [N,1], z_vals [N,S], xn [N*S,3], counts [N,2]); the gradient of xn reaches rot / trans."""

    @staticmethod
    def forward(ctx, rot, trans, fixed, owner, table, rows, noise, tables, rc, S, in_place):
        ctx.params = (rot, trans) if in_place else None
        rot, trans = _f32c(rot), _f32c(trans)
        fixed = _f32c(fixed) if fixed is not None and fixed.numel() else None
        owner = owner.to(torch.int64).contiguous()
        rows = rows.to(torch.int64).contiguous()
        F = 0 if fixed is None else fixed.shape[0]
        K, N, dev = rot.shape[0], rows.shape[0], rot.device
        flat = table.reshape(-1, 7)
        e = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)       # noqa: E731
        d_cam, rgb, depth, z_vals, xn = e(N, 3), e(N, 3), e(N, 1), e(N, S), e(N * S, 3)
        counts = torch.empty((N, 2), dtype=torch.int32, device=dev)
        zu, zoff, znd = tables
        with _timed("sample_rays"):
            check(lib().mipsf_gather_pose_place_fwd(dptr(flat), flat.shape[0], dptr(rows, torch.int64), dptr(fixed), dptr(rot),
                                                    dptr(trans), F, K, dptr(owner, torch.int64), dptr(noise), dptr(zu),
                                                    dptr(zoff), dptr(znd), C.byref(rc), dptr(d_cam), dptr(rgb), dptr(depth),
                                                    dptr(z_vals), dptr(xn), dptr(counts, torch.int32), N, stream_ptr()),
                  "gather_pose_place_fwd")
        ctx.F, ctx.K, ctx.N, ctx.S, ctx.rc = F, K, N, S, rc
        ctx.save_for_backward(rot, owner, d_cam, z_vals)
        ctx.mark_non_differentiable(rgb, depth, z_vals, counts)
        ctx.set_materialize_grads(False)
        return rgb, depth, z_vals, xn, counts

    @staticmethod
    def backward(ctx, _g_rgb, _g_depth, _g_z, dxn, _g_counts):
        if dxn is None:
            return (None,) * 11
        rot, owner, d_cam, z_vals = ctx.saved_tensors
        n = _lib.buffer_size(_lib.SIZE_PLACE_POSE_SCRATCH, ctx.N, ctx.F, ctx.K)
        key = (rot.device.index, stream_ptr(), "place", n)
        scratch = _POSE_SCRATCH.get(key)
        if scratch is None:
            scratch = _POSE_SCRATCH[key] = torch.zeros(n, dtype=torch.float32, device=rot.device)
        direct = ctx.params is not None and all(
            p.is_leaf and p.requires_grad and not p._backward_hooks and p.dtype == torch.float32 and p.is_contiguous()
            for p in ctx.params)
        if direct:
            for p in ctx.params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            d_rot, d_trans = ctx.params[0].grad, ctx.params[1].grad
        else:
            d_rot = torch.empty((ctx.K, 4), dtype=torch.float32, device=rot.device)
            d_trans = torch.empty((ctx.K, 3), dtype=torch.float32, device=rot.device)
        with _timed("rays_bwd"):
            check(lib().mipsf_place_pose_bwd(dptr(_f32c(dxn)), dptr(z_vals), C.byref(ctx.rc), dptr(rot), ctx.F, ctx.K,
                                             dptr(owner, torch.int64), dptr(d_cam), dptr(d_rot), dptr(d_trans), dptr(scratch),
                                             ctx.N, ctx.S, 1 if direct else 0, stream_ptr()), "place_pose_bwd")
        if direct:
            return (None,) * 11
        return (d_rot, d_trans) + (None,) * 9


# ----------------------------------------------------------------------------------- Adam
def adam_advance(step_dev, hyper_dev, lr, beta1, beta2):
    adam_advance_n([(step_dev, hyper_dev, lr, beta1, beta2)])


def adam_advance_n(groups):
    """groups: list of (step_dev int32[1], hyper_dev float[2], lr, beta1, beta2) -- one launch for all of them."""
    n = len(groups)
    steps = (C.c_void_p * n)(*[dptr(g[0], torch.int32) for g in groups])
    hypers = (C.c_void_p * n)(*[dptr(g[1]) for g in groups])
    lr = (C.c_float * n)(*[g[2] for g in groups])
    b1 = (C.c_float * n)(*[g[3] for g in groups])
    b2 = (C.c_float * n)(*[g[4] for g in groups])
    check(lib().mipsf_adam_advance_n(steps, hypers, lr, b1, b2, n, stream_ptr()), "adam_advance_n")


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, zero_grad=False,
              hyper_dev=None):
    with _timed("adam_step" if param.numel() > (1 << 20) else "adam_step_small"):
        check(lib().mipsf_adam_step(dptr(param), dptr(grad), dptr(exp_avg), dptr(exp_avg_sq), param.numel(), lr,
                                    beta1, beta2, eps, weight_decay, step, dptr(hyper_dev),
                                    1 if zero_grad else 0, stream_ptr()), "adam_step")


def adam_step_multi(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, weight_decay, step, zero_grad=False,
                    hyper_dev=None):
    """One launch for a group of (small) tensors sharing hyper-parameters and step count."""
    for i in range(0, len(params), _lib.ADAM_MAX_TENSORS):
        chunk = slice(i, i + _lib.ADAM_MAX_TENSORS)
        t = _lib.AdamTensors()
        t.count = len(params[chunk])
        for j, (p, g, m, v) in enumerate(zip(params[chunk], grads[chunk], exp_avgs[chunk], exp_avg_sqs[chunk])):
            t.param[j], t.grad[j], t.exp_avg[j], t.exp_avg_sq[j] = dptr(p), dptr(g), dptr(m), dptr(v)
            t.numel[j] = p.numel()
        with _timed("adam_step_small"):
            check(lib().mipsf_adam_step_multi(C.byref(t), lr, beta1, beta2, eps, weight_decay, step,
                                              dptr(hyper_dev), 1 if zero_grad else 0, stream_ptr()),
                  "adam_step_multi")


def adam_step_small(groups, zero_grad=False):
    """groups: list of (step_dev int32[1], hyper_dev float[2], lr, beta1, beta2, eps, weight_decay,
    [(param, grad, exp_avg, exp_avg_sq), ...]) with tiny tensors only -- the whole optimiser step in one launch."""
    d = _lib.AdamSmall()
    d.n_groups = len(groups)
    ti = 0
    for gi, (step_dev, hyper_dev, lr, b1, b2, eps, wd, tensors) in enumerate(groups):
        d.step_dev[gi], d.hyper_dev[gi] = dptr(step_dev, torch.int32), dptr(hyper_dev)
        d.lr[gi], d.beta1[gi], d.beta2[gi], d.eps[gi], d.weight_decay[gi] = lr, b1, b2, eps, wd
        for p, g, m, v in tensors:
            d.param[ti], d.grad[ti], d.exp_avg[ti], d.exp_avg_sq[ti] = dptr(p), dptr(g), dptr(m), dptr(v)
            d.numel[ti], d.group_of[ti] = p.numel(), gi
            ti += 1
    d.n_tensors = ti
    with _timed("adam_step_small"):
        check(lib().mipsf_adam_step_small(C.byref(d), 1 if zero_grad else 0, stream_ptr()), "adam_step_small")


def adam_step_all(groups, ticket, zero_grad=False):
    """groups as for ``adam_step_small`` but with tensors of any size: the whole optimiser step -- counters, bias corrections,
    every update -- in ONE launch (the map optimiser: hash table + the decoder's ten tensors).  ticket: an int32 tensor of
    MIPSF_ADAM_TICKET_WORDS (576) zeros owned by THIS optimiser (tickets + the scalars left for its next step)."""
    d = _lib.AdamSmall()
    d.n_groups = len(groups)
    ti, dev, big = 0, None, False
    for gi, (step_dev, hyper_dev, lr, b1, b2, eps, wd, tensors) in enumerate(groups):
        d.step_dev[gi], d.hyper_dev[gi] = dptr(step_dev, torch.int32), dptr(hyper_dev)
        d.lr[gi], d.beta1[gi], d.beta2[gi], d.eps[gi], d.weight_decay[gi] = lr, b1, b2, eps, wd
        for p, g, m, v in tensors:
            d.param[ti], d.grad[ti], d.exp_avg[ti], d.exp_avg_sq[ti] = dptr(p), dptr(g), dptr(m), dptr(v)
            d.numel[ti], d.group_of[ti] = p.numel(), gi
            dev, big = p.device, big or p.numel() > (1 << 20)
            ti += 1
    d.n_tensors = ti
    with _timed("adam_step" if big else "adam_step_small"):
        check(lib().mipsf_adam_step_all(C.byref(d), 1 if zero_grad else 0, dptr(ticket, torch.int32), stream_ptr()),
              "adam_step_all")


def ro_fitness(raw, target_d, trunc: float, point_major: bool = False) -> torch.Tensor:
    """raw [P,n,10] (run_network output) or [P,n,1] (SDF only; with point_major the memory order is [n,P] and raw is
    passed as sdf.view(P, n, 1) all the same), target_d [n] -> mean_masked_sdf [P] (RandomOptimizer.py:125-129)."""
    P, n, stride = raw.shape
    out = torch.empty(P, dtype=torch.float32, device=raw.device)
    if stride == 1:
        check(lib().mipsf_ro_fitness_sdf(dptr(raw), dptr(target_d), trunc, dptr(out), P, n, 1 if point_major else 0,
                                         stream_ptr()), "ro_fitness")
    else:
        check(lib().mipsf_ro_fitness(dptr(raw), stride, dptr(target_d), trunc, dptr(out), P, n, stream_ptr()),
              "ro_fitness")
    return out


RO_STATE_FLOATS = 32


def ro_particles(pst, state, rays_d_cam, target_d, rc, point_major: bool = False):
    """One RandomOptimizer round, first half (RandomOptimizer.py:184-190, 117-121): -> xn [P*n,3], pst7 [P,7].
    point_major: sample (particle p, lattice point j) is row j*P + p instead of p*n + j."""
    P, n = pst.shape[0], rays_d_cam.shape[0]
    xn = torch.empty((P * n, 3), dtype=torch.float32, device=pst.device)
    pst7 = torch.empty((P, 7), dtype=torch.float32, device=pst.device)
    with _timed("ro_particles"):
        check(lib().mipsf_ro_particles(dptr(pst), dptr(state), dptr(rays_d_cam), dptr(target_d), C.byref(rc), dptr(xn),
                                       dptr(pst7), P, n, 1 if point_major else 0, stream_ptr()), "ro_particles")
    return xn, pst7


def ro_update(mean_masked, pst7, state, sdf_weight: float, rescale: float):
    """Second half (RandomOptimizer.py:196-224): updates ``state`` (rot, trans, search size) in place on the device."""
    with _timed("ro_update"):
        check(lib().mipsf_ro_update(dptr(mean_masked), dptr(pst7), dptr(state), sdf_weight, rescale,
                                    mean_masked.shape[0], stream_ptr()), "ro_update")


def gather_rays(db, idx, split: bool = False):
    """db [..., 7] device-resident ray rows, idx int64 [N] (device) -> rays [N,7], or (d_cam, rgb, depth[N,1])."""
    flat = db.reshape(-1, 7)
    N = idx.shape[0]
    dev = flat.device
    if split:
        d_cam = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rgb = torch.empty((N, 3), dtype=torch.float32, device=dev)
        depth = torch.empty((N, 1), dtype=torch.float32, device=dev)
        check(lib().mipsf_gather_rays(dptr(flat), flat.shape[0], dptr(idx, torch.int64), N, None, dptr(d_cam),
                                      dptr(rgb), dptr(depth), stream_ptr()), "gather_rays")
        return d_cam, rgb, depth
    out = torch.empty((N, 7), dtype=torch.float32, device=dev)
    check(lib().mipsf_gather_rays(dptr(flat), flat.shape[0], dptr(idx, torch.int64), N, dptr(out), None, None, None,
                                  stream_ptr()), "gather_rays")
    return out
