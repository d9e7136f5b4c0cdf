"""ctypes binding of libmipsf_hip.so (C ABI: include/mipsf.h).

The product path has no fallback: a missing library raises at first use, and every
operator raises when its tensors are not on a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIPSF_LIB (developer switch): an experiment build of the same C ABI (tools/micro/variant.sh) instead of the in-tree library
LIB_PATH = os.environ.get("MIPSF_LIB") or os.path.join(_HERE, "libmipsf_hip.so")
MAX_LEVELS = 32
FEAT_AOS, FEAT_LEVEL_MAJOR = 0, 1
PREC = {"f32": 0, "f16x3": 1, "f16": 2, "bf16x3": 3, "bf16x6": 4}      # MIPSF_PREC_* of include/mipsf.h


class GridMeta(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("log2_hashmap_size", C.c_uint32),
                ("base_resolution", C.c_uint32), ("per_level_scale", C.c_float),
                ("log2_per_level_scale", C.c_float), ("n_params", C.c_uint32),
                ("offsets", C.c_uint32 * (MAX_LEVELS + 1)), ("resolutions", C.c_uint32 * MAX_LEVELS),
                ("scales", C.c_float * MAX_LEVELS)]


class DecoderWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("w_pts0", "b_pts0", "w_pts2", "b_pts2", "w_rgb0", "b_rgb0",
                                          "w_sdf0", "b_sdf0", "w_sdf2", "b_sdf2")]


DecoderGrads = DecoderWeights   # same field order, non-const pointers


class RenderCfg(C.Structure):
    _fields_ = [("n_uniform", C.c_uint32), ("n_near", C.c_uint32), ("perturb", C.c_int), ("use_bound", C.c_int),
                ("bound_min", C.c_double * 3), ("bound_max", C.c_double * 3), ("half_len", C.c_double * 3),
                ("norm_factor", C.c_double), ("trunc", C.c_float), ("sc_factor", C.c_float),
                ("depth_trunc", C.c_float), ("rgb_missing_nonzero", C.c_int), ("emd_w", C.c_float)]


def _args(name, fields):
    """ctypes mirror of an argument block of include/mipsf.h (struct_size first; `new()` returns a zeroed, sized block)."""
    cls = type(name, (C.Structure,), {"_fields_": [("struct_size", C.c_uint32)] + fields})

    def new(**kw):
        a = cls()
        a.struct_size = C.sizeof(cls)
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    cls.new = staticmethod(new)
    return cls


_VP, _CU, _CI = C.c_void_p, C.c_uint32, C.c_int
HashgridBwdArgs = _args("HashgridBwdArgs", [("M", _CU), ("x", _VP), ("params", _VP), ("dout", _VP), ("dparams", _VP), ("dx", _VP),
                                            ("scratch", _VP), ("counters", _VP), ("meta", C.POINTER(GridMeta)),
                                            ("feat_layout", _CI), ("flags", _CU)])
DecoderFwd16Args = _args("DecoderFwd16Args", [("M", _CU), ("packed16", _VP), ("feat", _VP), ("x", _VP), ("out", _VP), ("saved", _VP),
                                              ("tile_live_clear", _VP), ("feat_layout", _CI), ("precision", _CI),
                                              ("sdf_only", _CI), ("lean_record", _CI), ("packed16_floats", _CU)])
DecoderChain16Args = _args("DecoderChain16Args", [("M", _CU), ("packed16", _VP), ("x", _VP), ("out", _VP), ("dout", _VP),
                                                  ("saved", _VP), ("dfeat", _VP), ("dx", _VP), ("dact", _VP), ("tile_live", _VP),
                                                  ("feat_layout", _CI), ("flags", _CI), ("packed16_floats", _CU)])
DecoderWgrad16Args = _args("DecoderWgrad16Args", [("M", _CU), ("packed16", _VP), ("feat", _VP), ("x", _VP), ("saved", _VP),
                                                  ("dact", _VP), ("tile_live", _VP), ("grads", C.POINTER(DecoderGrads)),
                                                  ("partial", _VP), ("feat_layout", _CI), ("arithmetic", _CI), ("flags", _CU),
                                                  ("packed16_floats", _CU)])
RenderFwdArgs = _args("RenderFwdArgs", [("N", _CU), ("S", _CU), ("raw", _VP), ("z_vals", _VP), ("target_rgb", _VP), ("target_d", _VP),
                                        ("counts", _VP), ("cfg", C.POINTER(RenderCfg)), ("rgb", _VP), ("depth", _VP),
                                        ("depth_var", _VP), ("disp", _VP), ("acc", _VP), ("weights", _VP), ("losses", _VP),
                                        ("partial", _VP), ("loss_weights", _VP), ("loss_total", _VP), ("ticket", _VP),
                                        ("sums", _VP), ("draw", _VP)])
RenderBwdArgs = _args("RenderBwdArgs", [("N", _CU), ("S", _CU), ("N_norm", _CU), ("raw", _VP), ("z_vals", _VP), ("target_rgb", _VP),
                                        ("target_d", _VP), ("counts", _VP), ("losses", _VP), ("cfg", C.POINTER(RenderCfg)),
                                        ("g_losses", _VP), ("g_total", _VP), ("loss_weights", _VP), ("g_rgb", _VP),
                                        ("g_depth", _VP), ("draw", _VP), ("flags", _CU)])
RENDER_BWD_KEEP_IF_UNIT = 1

# mipsf_buffer_size(which, n, a, b, meta): MIPSF_SIZE_* of include/mipsf.h
(SIZE_HASHGRID_BWD_SCRATCH, SIZE_HASHGRID_COUNTER_WORDS, SIZE_DECODER_PACKED, SIZE_DECODER_SAVED, SIZE_DECODER_DACT,
 SIZE_DECODER_WGRAD_PARTIAL, SIZE_DECODER_PACKED16, SIZE_DECODER_TILE_WORDS, SIZE_RENDER_PARTIAL, SIZE_PLACE_POSE_SCRATCH,
 SIZE_POSE_RAYS_SCRATCH) = range(1, 12)


ADAM_MAX_TENSORS = 16


class AdamTensors(C.Structure):
    _fields_ = [("count", C.c_uint32), ("param", C.c_void_p * ADAM_MAX_TENSORS), ("grad", C.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS),
                ("numel", C.c_uint64 * ADAM_MAX_TENSORS)]


ADAM_MAX_GROUPS = 8
ADAM_SMALL_MAX_NUMEL = 16384


class AdamSmall(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("n_tensors", C.c_uint32),
                ("step_dev", C.c_void_p * ADAM_MAX_GROUPS), ("hyper_dev", C.c_void_p * ADAM_MAX_GROUPS),
                ("lr", C.c_float * ADAM_MAX_GROUPS), ("beta1", C.c_float * ADAM_MAX_GROUPS),
                ("beta2", C.c_float * ADAM_MAX_GROUPS), ("eps", C.c_float * ADAM_MAX_GROUPS),
                ("weight_decay", C.c_float * ADAM_MAX_GROUPS),
                ("param", C.c_void_p * ADAM_MAX_TENSORS), ("grad", C.c_void_p * ADAM_MAX_TENSORS),
                ("exp_avg", C.c_void_p * ADAM_MAX_TENSORS), ("exp_avg_sq", C.c_void_p * ADAM_MAX_TENSORS),
                ("numel", C.c_uint32 * ADAM_MAX_TENSORS), ("group_of", C.c_uint32 * ADAM_MAX_TENSORS)]


_P = C.c_void_p
_U32, _U64, _I, _F, _D = C.c_uint32, C.c_uint64, C.c_int, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol include/mipsf.h declares
SIGNATURES = {
    "mipsf_last_error": (C.c_char_p, []),
    "mipsf_abi_version": (_I, []),
    "mipsf_device_cu_count": (_I, []),
    "mipsf_buffer_size": (_U64, [_I, _U32, _U32, _U32, C.POINTER(GridMeta)]),
    "mipsf_hashgrid_meta_init": (_I, [C.POINTER(GridMeta), _U32, _U32, _U32, _U32, _D]),
    "mipsf_hashgrid_fwd": (_I, [_P, _P, _P, _P, _U32, C.POINTER(GridMeta), _I, _P]),
    "mipsf_hashgrid_dx_from_jac": (_I, [_P, _P, _P, _P, _U32, C.POINTER(GridMeta), _I, _P]),
    "mipsf_hashgrid_bwd": (_I, [C.POINTER(HashgridBwdArgs), _P]),
    "mipsf_hashgrid_route": (_I, [_P, _P, _U32, C.POINTER(GridMeta), _P]),
    "mipsf_hashgrid_indices": (_I, [_P, _P, _U32, C.POINTER(GridMeta), _P]),
    "mipsf_freq_fwd": (_I, [_P, _P, _U32, _U32, _U32, _P]),
    "mipsf_freq_bwd": (_I, [_P, _P, _P, _U32, _U32, _U32, _P]),
    "mipsf_decoder_pack": (_I, [C.POINTER(DecoderWeights), _P, _P]),
    "mipsf_decoder_pack_host": (_I, [C.POINTER(DecoderWeights), _P]),
    "mipsf_decoder_fwd": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _U32, _P]),
    "mipsf_decoder_fwd_sdf": (_I, [_P, _P, _I, _P, _P, _I, _P, _U32, _P]),
    "mipsf_decoder_bwd": (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, C.POINTER(DecoderGrads), _P, _P,
                               _U32, _P]),
    "mipsf_decoder_bwd_chain": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _U32, _P]),
    "mipsf_decoder_wgrad": (_I, [_P, _I, _P, _P, _I, _P, _P, C.POINTER(DecoderGrads), _P, _I, _U32, _P]),
    "mipsf_decoder_pack16": (_I, [C.POINTER(DecoderWeights), _P, _I, _P]),
    "mipsf_decoder_fwd16": (_I, [C.POINTER(DecoderFwd16Args), _P]),
    "mipsf_decoder_bwd_chain16": (_I, [C.POINTER(DecoderChain16Args), _P]),
    "mipsf_decoder_wgrad16": (_I, [C.POINTER(DecoderWgrad16Args), _P]),
    "mipsf_sample_rays": (_I, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(RenderCfg), _P, _P, _P, _U32, _P]),
    "mipsf_normalise_points": (_I, [_P, C.POINTER(RenderCfg), _P, _U32, _P]),
    "mipsf_render_fwd": (_I, [C.POINTER(RenderFwdArgs), _P]),
    "mipsf_loss_finalize_sums": (_I, [_P, C.POINTER(RenderCfg), _U32, _U32, _P, _P, _P, _P]),
    "mipsf_render_bwd": (_I, [C.POINTER(RenderBwdArgs), _P]),
    "mipsf_gather_pose_place_fwd": (_I, [_P, _U64, _P, _P, _P, _P, _U32, _U32, _P, _P, _P, _P, _P, C.POINTER(RenderCfg), _P, _P,
                                         _P, _P, _P, _P, _U32, _P]),
    "mipsf_place_pose_bwd": (_I, [_P, _P, C.POINTER(RenderCfg), _P, _U32, _U32, _P, _P, _P, _P, _P, _U32, _U32, _I, _P]),
    "mipsf_rays_bwd": (_I, [_P, _P, C.POINTER(RenderCfg), _P, _P, _U32, _U32, _P]),
    "mipsf_normalise_bwd": (_I, [_P, C.POINTER(RenderCfg), _P, _U32, _P]),
    "mipsf_pose_rays_fwd": (_I, [_P, _U64, _P, _P, _P, _P, _U32, _U32, _P, _P, _P, _P, _P, _P, _U32, _P]),
    "mipsf_pose_handover": (_I, [_P, _I, _P, _P, _P]),
    "mipsf_pose_rays_bwd": (_I, [_P, _P, _P, _U32, _U32, _P, _P, _P, _P, _P, _U32, _I, _P]),
    "mipsf_adam_step": (_I, [_P, _P, _P, _P, _U64, _F, _F, _F, _F, _F, _U32, _P, _I, _P]),
    "mipsf_adam_advance_n": (_I, [_P, _P, _P, _P, _P, _U32, _P]),
    "mipsf_adam_step_multi": (_I, [C.POINTER(AdamTensors), _F, _F, _F, _F, _F, _U32, _P, _I, _P]),
    "mipsf_adam_step_small": (_I, [C.POINTER(AdamSmall), _I, _P]),
    "mipsf_adam_step_all": (_I, [C.POINTER(AdamSmall), _I, _P, _P]),
    "mipsf_ro_fitness": (_I, [_P, _U32, _P, _F, _P, _U32, _U32, _P]),
    "mipsf_ro_fitness_sdf": (_I, [_P, _P, _F, _P, _U32, _U32, _I, _P]),
    "mipsf_ro_particles": (_I, [_P, _P, _P, _P, C.POINTER(RenderCfg), _P, _P, _U32, _U32, _I, _P]),
    "mipsf_ro_update": (_I, [_P, _P, _P, _F, _F, _U32, _P]),
    "mipsf_gather_rays": (_I, [_P, _U64, _P, _U32, _P, _P, _P, _P, _P]),
}


def buffer_size(which: int, n: int = 0, a: int = 0, b: int = 0, meta=None) -> int:
    """mipsf_buffer_size: elements of a scratch / record buffer (SIZE_* above)."""
    v = lib().mipsf_buffer_size(which, n, a, b, C.byref(meta) if meta is not None else None)
    if v == 0xFFFFFFFFFFFFFFFF:
        raise RuntimeError((lib().mipsf_last_error() or b"mipsf_buffer_size: bad query").decode())
    return int(v)


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mipsfusion_amd/csrc`). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.mipsf_abi_version() != 2:
            raise RuntimeError("libmipsf_hip.so ABI version mismatch")
        _lib = handle
    return _lib


# Caches of caller-kept counter / ticket blocks (ops._scatter_counters, ops._zeroed_words, ops._pose_scratch, FusedAdam's
# ticket): a kernel finds them at zero and leaves them at zero -- IF it runs to completion.  A call that failed after launching
# part of its work may have left counts or tickets behind; every cache registered here is emptied when a call fails, so the
# next call starts from freshly zeroed blocks instead of silently miscounting.
KEPT_BLOCK_CACHES = []


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().mipsf_last_error().decode(errors="replace")
        for cache in KEPT_BLOCK_CACHES:
            cache.clear()
        raise RuntimeError(f"libmipsf_hip {what} failed (code {rc}): {msg}")


def stream_ptr() -> int:
    """hipStream_t of torch's current stream on the current device (raw query: 0.2 us instead of 2.6 us for
    torch.cuda.current_stream().cuda_stream -- it is called once per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def dptr(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    """Device pointer of a contiguous GPU tensor (None passes through as NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("mipsfusion_amd operators need GPU tensors (no CPU fallback exists)")
    if t.device.index != torch._C._cuda_getDevice():
        # launches go to the CURRENT device's current stream (stream_ptr): a tensor of another GPU would be a foreign
        # pointer there.  One process per GPU with torch.cuda.set_device(local_rank) is the supported arrangement.
        raise RuntimeError(f"tensor lives on cuda:{t.device.index} but the current device is "
                           f"cuda:{torch._C._cuda_getDevice()}; wrap the call in torch.cuda.device(tensor.device)")
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return t.data_ptr()


def make_grid_meta(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=16,
                   per_level_scale=2.0) -> GridMeta:
    m = GridMeta()
    check(lib().mipsf_hashgrid_meta_init(C.byref(m), n_levels, n_features, log2_hashmap_size, base_resolution,
                                         float(per_level_scale)), "hashgrid_meta_init")
    return m
