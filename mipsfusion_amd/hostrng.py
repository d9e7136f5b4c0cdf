"""torch's default CPU generator, replicated on the host in C (csrc/host/hostrng.c) for the two draws the reference's
samplers make per iteration -- ``torch.randn_like(depth)`` (sampling_helper.py:30, :62) and ``torch.rand(N, S)``
(scene_rep.py:176).  Same stream, same values to the last bit, ~4x faster: the generator stage is what bounds the frame
time when the index stream has to be the reference's own (mipsfusion_amd/sequence.py).

    with hostrng.session() as g:          # takes the generator state from torch ...
        g.randn_(scores)                  # ... fills contiguous float32 CPU tensors in place ...
        g.rand_(noise)
                                          # ... and hands the advanced state back to torch

The replica is CHECKED against torch on first use (values of both draws incl. a ragged tail, and the generator state
afterwards); if this build of torch should differ, or the library is missing, the same calls run torch's own functions.
Not thread-safe against other users of the default generator while a session is open (neither is the generator's stream
order, which is the point of running it on one thread)."""
import contextlib
import threading
import ctypes as C
import os
import random

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmipsf_hostrng.so")
_N = 624
# torch.get_rng_state() of the CPU generator: CPUGeneratorImplState = {seed u64 @0, left i32 @8, seeded i32 @12,
# next u64 @16, state u64[624] @24, normal cache ...} (5056 bytes)
_OFF_LEFT, _OFF_NEXT, _OFF_STATE, _STATE_BYTES = 8, 16, 24, 5056


class _MT(C.Structure):
    _fields_ = [("state", C.c_uint32 * _N), ("left", C.c_int32), ("next", C.c_uint32)]


_lib = None
_ok = None
_topk_ok = None
_py_ok = None


def _load():
    global _lib
    if _lib is None and os.path.exists(_LIB_PATH):
        lib = C.CDLL(_LIB_PATH)
        lib.mipsf_mt_uniform_f32.argtypes = [C.POINTER(_MT), C.c_void_p, C.c_int64]
        lib.mipsf_mt_uniform_f32.restype = None
        lib.mipsf_mt_normal_f32.argtypes = [C.POINTER(_MT), C.c_void_p, C.c_int64, C.c_int]
        lib.mipsf_mt_normal_f32.restype = C.c_int
        lib.mipsf_topk_valid_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                C.c_void_p]
        lib.mipsf_topk_valid_scores.restype = C.c_int
        lib.mipsf_py_sample_range.argtypes = [C.POINTER(_MT), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        lib.mipsf_py_sample_range.restype = C.c_int
        _lib = lib
    return _lib


def _guarded(check) -> bool:
    """a replica is used only if the library loads, is not switched off, and its check against the library routine passes
    -- a check that cannot even run (another generator state layout, a missing symbol) counts as failed"""
    if os.environ.get("MIPSF_NO_HOSTRNG"):
        return False
    try:
        return bool(_load()) and bool(check())
    except Exception:                                            # noqa: BLE001 -- any failure selects the library routine
        return False


class _Session:
    """the generator state lives here between `take` and `give`"""

    def __init__(self, native: bool, threads: int):
        self.native, self.threads = native, threads
        self.mt = _MT()
        self.raw = None

    def take(self):
        if not self.native:
            return
        self.raw = torch.get_rng_state()
        b = self.raw.numpy()
        if b.shape[0] != _STATE_BYTES:
            raise RuntimeError("unexpected CPU generator state size %d" % b.shape[0])
        np.ctypeslib.as_array(self.mt.state)[:] = b[_OFF_STATE:_OFF_STATE + 8 * _N].view(np.uint64).astype(np.uint32)
        self.mt.left = int(b[_OFF_LEFT:_OFF_LEFT + 4].view(np.int32)[0])
        self.mt.next = int(b[_OFF_NEXT:_OFF_NEXT + 8].view(np.uint64)[0])

    def give(self):
        if not self.native:
            return
        b = self.raw.numpy()
        b[_OFF_STATE:_OFF_STATE + 8 * _N].view(np.uint64)[:] = np.ctypeslib.as_array(self.mt.state).astype(np.uint64)
        b[_OFF_LEFT:_OFF_LEFT + 4].view(np.int32)[0] = self.mt.left
        b[_OFF_NEXT:_OFF_NEXT + 8].view(np.uint64)[0] = self.mt.next
        torch.set_rng_state(self.raw)

    @staticmethod
    def _check(t):
        if t.device.type != "cpu" or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("host RNG fills contiguous float32 CPU tensors")

    def through_torch(self, fn):
        """run `fn` (anything that draws from torch's default CPU generator itself) inside the session: the state goes
        back to torch for the call and is taken again afterwards"""
        self.give()
        try:
            return fn()
        finally:
            self.take()

    def randn_(self, out: torch.Tensor) -> torch.Tensor:
        """out.normal_() of the default generator"""
        if not self.native or out.numel() < 16:     # (torch's scalar path for tiny tensors keeps a cached second value)
            return self.through_torch(out.normal_)
        self._check(out)
        _lib.mipsf_mt_normal_f32(C.byref(self.mt), out.data_ptr(), out.numel(), self.threads)
        return out

    def rand_(self, out: torch.Tensor) -> torch.Tensor:
        """out.uniform_() of the default generator"""
        if not self.native:
            return out.uniform_()
        self._check(out)
        _lib.mipsf_mt_uniform_f32(C.byref(self.mt), out.data_ptr(), out.numel())
        return out


def _self_check(threads: int) -> bool:
    keep = torch.get_rng_state()
    try:
        for seed, n in ((20240917, 4096 + 5), (7, 16), (99, 285200)):
            torch.manual_seed(seed)
            ref_n, ref_u = torch.empty(n).normal_(), torch.empty(1000).uniform_()
            ref_state = torch.get_rng_state()
            torch.manual_seed(seed)
            s = _Session(True, threads)
            s.take()
            got_n, got_u = s.randn_(torch.empty(n)), s.rand_(torch.empty(1000))
            s.give()
            if not (torch.equal(got_n, ref_n) and torch.equal(got_u, ref_u) and torch.equal(torch.get_rng_state(), ref_state)):
                return False
        return True
    finally:
        torch.set_rng_state(keep)


def available(threads: int = 4) -> bool:
    """True when the C replica is present AND reproduces this torch build's draws bit for bit (checked once)."""
    global _ok
    if _ok is None:
        _ok = _guarded(lambda: _self_check(threads))
    return _ok


# One session at a time, process wide: a session takes the default generator's state out of torch, draws, and puts it
# back -- a second thread doing the same in between (the frame loop's sample producer, a caller's own worker) would have
# its draws overwritten or duplicated.  (Re-entrant only so that a mistaken nested session cannot deadlock its own thread; a
# nested session's draws are lost when the outer one puts ITS state back -- draw through the session you hold.)
# CONTRACT for callers with a producer thread (sequence.py's sample producer holds a session for a whole frame plan, including
# bounded-queue puts that wait for the top-k stage): another thread that opens a session meanwhile WAITS for that plan to
# finish -- milliseconds, never a deadlock as long as the stage that drains the queue opens no session itself (it does not:
# the top-k half is generator-free).  That wait is the point: the generator's stream is ONE sequence in the reference's call
# order, and a draw squeezed in between the producer's would change every index after it.  Code that needs draws in a fixed
# place of the stream asks the producer for them (the frame plan) instead of opening its own session.
_SESSION_LOCK = threading.RLock()


@contextlib.contextmanager
def session(threads: int = 4):
    with _SESSION_LOCK:
        s = _Session(available(threads), threads)
        s.take()
        try:
            yield s
        finally:
            s.give()


# ---------------------------------------------------------------------------------------------- scores + top-k
def _topk_torch(depth, draw, k, blocked):
    """sampling_helper._valid_scores + torch.topk, as the reference computes them (sampling_helper.py:24-33, :62-66)"""
    valid = (depth.flatten() > 0.).to(depth.dtype)
    if blocked is not None:
        valid = valid * (1 - blocked.to(depth.dtype))
    return torch.topk(valid * torch.abs(draw), k)[1]


def _topk_native(depth, draw, k, blocked, out=None):
    n = depth.numel()
    for t in (depth, draw):
        if t.device.type != "cpu" or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            return None
    if blocked is not None and (blocked.dtype != torch.uint8 or not blocked.is_contiguous() or blocked.numel() != n):
        return None
    if out is None:
        out = torch.empty(k, dtype=torch.int64)
    scratch = np.empty(k, dtype=np.int64)
    rc = _lib.mipsf_topk_valid_scores(depth.data_ptr(), draw.data_ptr(), blocked.data_ptr() if blocked is not None else None,
                                      n, k, out.data_ptr(), scratch.ctypes.data)
    return out if rc == 0 else None


def _topk_self_check() -> bool:
    g = torch.Generator().manual_seed(5)
    for n, k, p_valid, with_block in ((70000, 1000, 0.95, True), (70000, 1000, 0.01, True), (8192, 128, 0.5, False),
                                      (8192, 128, 0.0, False), (285200, 2048, 0.9, True)):
        depth = torch.rand(n, generator=g) * (torch.rand(n, generator=g) < p_valid)
        draw = torch.randn(n, generator=g)
        m = draw[3::7].numel()
        draw[0::7][:m] = -draw[3::7]                                        # exact ties among the valid scores, too
        blocked = (torch.rand(n, generator=g) < 0.01).to(torch.uint8) if with_block else None
        want = _topk_torch(depth, draw, k, blocked)
        for d in (draw, torch.randn(n, generator=g)):                       # with ties / (almost surely) without
            if d is not draw:
                want = _topk_torch(depth, d, k, blocked)
            got = _topk_native(depth, d, k, blocked)
            if got is None or not torch.equal(got, want):
                return False
    return True


def topk_available() -> bool:
    """True when the one-pass score + top-k of the library returns torch.topk's indices in torch.topk's order here."""
    global _topk_ok
    if _topk_ok is None:
        _topk_ok = _guarded(_topk_self_check)
    return _topk_ok


def topk_valid_pixels(depth: torch.Tensor, draw: torch.Tensor, k: int, blocked: torch.Tensor = None) -> torch.Tensor:
    """``torch.topk((depth > 0) * |draw|, k)[1]`` with the pixels of the uint8 mask `blocked` scored 0 -- the second half
    of sample_valid_pixels_random / sample_pixels_mix (sampling_helper.py:24-33, :55-68) -- in one pass, same indices in
    the same order (ties included); torch's own ops when the library is missing, disagrees, or k * 64 > n."""
    if topk_available():
        got = _topk_native(depth, draw, k, blocked)
        if got is not None:
            return got
    return _topk_torch(depth, draw, k, blocked)


# ---------------------------------------------------------------------------------------------- python's `random`
class _PySession:
    """python's global `random` generator between `take` and `give`: ``sample_range(n, k)`` == ``torch.tensor(
    random.sample(range(n), k))`` (keyframeSet.py:386-436 draws the rays of a mapping iteration this way, ~1 us per
    index in the interpreter).  As with the torch session: whoever else draws from the global generator while a session
    is open sees the state of `take` time and is overwritten by `give` -- one thread owns a generator's stream."""

    def __init__(self, native: bool):
        self.native = native
        self.mt = _MT()
        self._version, self._gauss = 3, None
        self._scratch = np.empty(0, dtype=np.int64)

    def take(self):
        if not self.native:
            return
        self._version, internal, self._gauss = random.getstate()
        np.ctypeslib.as_array(self.mt.state)[:] = np.array(internal[:_N], dtype=np.uint32)
        self.mt.next = internal[_N]

    def give(self):
        if not self.native:
            return
        random.setstate((self._version, tuple(np.ctypeslib.as_array(self.mt.state).tolist()) + (int(self.mt.next),),
                         self._gauss))

    def sample_range(self, n: int, k: int) -> torch.Tensor:
        if self.native:
            out = torch.empty(k, dtype=torch.int64)
            if self._scratch.shape[0] < n:
                self._scratch = np.empty(n, dtype=np.int64)
            if _lib.mipsf_py_sample_range(C.byref(self.mt), n, k, out.data_ptr(), self._scratch.ctypes.data) == 0:
                return out
            self.give()                         # arguments the replica refuses: python's own function (and its errors)
            try:
                return torch.tensor(random.sample(range(n), k))
            finally:
                self.take()
        return torch.tensor(random.sample(range(n), k))


def _py_self_check() -> bool:
    keep = random.getstate()
    try:
        for seed, n, k in ((1, 5000, 409), (2, 35000, 2048), (3, 600, 500), (4, 20, 5), (5, 1, 1), (6, 5000, 0),
                           (7, 4096, 1365), (8, 87, 22)):
            random.seed(seed)
            random.random()                                 # (an odd position inside the state block)
            want = [random.sample(range(n), k) for _ in range(3)]
            end = random.getstate()
            random.seed(seed)
            random.random()
            s = _PySession(True)
            s.take()
            got = [s.sample_range(n, k).tolist() for _ in range(3)]
            s.give()
            if got != want or random.getstate() != end:
                return False
        return True
    finally:
        random.setstate(keep)


def py_available() -> bool:
    global _py_ok
    if _py_ok is None:
        _py_ok = _guarded(_py_self_check)
    return _py_ok


@contextlib.contextmanager
def py_session():
    s = _PySession(py_available())
    s.take()
    try:
        yield s
    finally:
        s.give()
