"""ctypes access to oracle/c/libhashgrid_ref.so (plain-C hash-grid restatement).  TEST INFRASTRUCTURE."""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c", "libhashgrid_ref.so")


class _Desc(C.Structure):
    _fields_ = [("n_levels", C.c_uint32), ("n_features", C.c_uint32), ("offsets", C.c_void_p),
                ("resolutions", C.c_void_p), ("scales", C.c_void_p)]


def available():
    return os.path.exists(_SO)


def hashgrid(x: np.ndarray, params, meta):
    """x [M,3] fp32, meta: oracle.tcnn_cpu.GridMeta -> (idx [M,L,8] uint32, y [M,2L] fp32 or None)."""
    lib = C.CDLL(_SO)
    x = np.ascontiguousarray(x, dtype=np.float32)
    offs = np.array(meta.offsets, dtype=np.uint32)
    res = np.array(meta.resolutions, dtype=np.uint32)
    sc = np.array(meta.scales, dtype=np.float32)
    d = _Desc(meta.n_levels, meta.n_features, offs.ctypes.data, res.ctypes.data, sc.ctypes.data)
    M = x.shape[0]
    idx = np.zeros((M, meta.n_levels, 8), dtype=np.uint32)
    y = None
    pp = None
    if params is not None:
        params = np.ascontiguousarray(params, dtype=np.float32)
        y = np.zeros((M, meta.n_levels * 2), dtype=np.float32)
        pp = params.ctypes.data
    lib.hashgrid_ref.restype = None
    lib.hashgrid_ref(C.c_void_p(x.ctypes.data), C.c_void_p(pp), C.byref(d), C.c_uint32(M),
                     C.c_void_p(idx.ctypes.data), C.c_void_p(y.ctypes.data if y is not None else None))
    return idx, y
