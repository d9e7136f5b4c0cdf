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
import ctypes as C
import os

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


def _load():
    global _lib
    if _lib is None and os.path.exists(_LIB_PATH):
        lib = C.CDLL(_LIB_PATH)
        lib.mipsf_mt_uniform_f32.argtypes = [C.POINTER(_MT), C.c_void_p, C.c_int64]
        lib.mipsf_mt_uniform_f32.restype = None
        lib.mipsf_mt_normal_f32.argtypes = [C.POINTER(_MT), C.c_void_p, C.c_int64, C.c_int]
        lib.mipsf_mt_normal_f32.restype = C.c_int
        _lib = lib
    return _lib


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
        _ok = bool(_load()) and not os.environ.get("MIPSF_NO_HOSTRNG") and _self_check(threads)
    return _ok


@contextlib.contextmanager
def session(threads: int = 4):
    s = _Session(available(threads), threads)
    s.take()
    try:
        yield s
    finally:
        s.give()
