"""ctypes loader of the C restatement (oracle/spmm_ref.c) -- TEST INFRASTRUCTURE."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_spmm.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def spmm_f32_chunked(rowptr, col, val, x, chunk, groups=1, long_row=-1):
    """fp32 SpMM in the HIP kernel's association order (bit-exact comparison target).  groups / long_row: the fix-up's
    order for rows of more than long_row pieces after the first (lpg_spmm_fixup_groups / lpg_spmm_fixup_long_row)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = None if val is None else np.ascontiguousarray(val, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = len(rowptr) - 1, x.shape[1]
    y = np.empty((n, d), dtype=np.float32)
    load().lpg_oracle_spmm_f32_chunked2(_p(rowptr), _p(col), _p(val), _p(x), ctypes.c_int64(n), ctypes.c_int64(d),
                                        ctypes.c_int64(chunk), ctypes.c_int64(groups), ctypes.c_int64(long_row), _p(y))
    return y


def scatter_add_f32(src, dst, w, x, n_rows):
    src = np.ascontiguousarray(src, dtype=np.int64)
    dst = np.ascontiguousarray(dst, dtype=np.int64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty((n_rows, x.shape[1]), dtype=np.float32)
    load().lpg_oracle_scatter_add_f32(_p(src), _p(dst), _p(w), _p(x), ctypes.c_int64(len(src)), ctypes.c_int64(n_rows),
                                      ctypes.c_int64(x.shape[1]), _p(y))
    return y
