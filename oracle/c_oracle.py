"""ctypes wrapper of oracle/libwmf_oracle.so (the C restatement; test infrastructure only)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libwmf_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libwmf_oracle.so not built: run `make -C oracle`")
        _lib = ctypes.CDLL(path)
        _lib.wmf_oracle_half_step.restype = ctypes.c_int
    return _lib


def set_threads(n):
    """OpenMP threads for the following calls; returns the number in effect."""
    return int(_load().wmf_oracle_set_threads(ctypes.c_int(int(n))))


def half_step(Y, C, lam, bias=False):
    """float64 [n, f] result of recompute_factors[_bias](Y, C, lam) computed in double (all host cores)."""
    lib = _load()
    Y = np.ascontiguousarray(Y, dtype=np.float32)
    indptr = np.ascontiguousarray(C.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(C.indices, dtype=np.int32)
    values = np.ascontiguousarray(C.data, dtype=np.float64)
    n, f = C.shape[0], Y.shape[1]
    X = np.zeros((n, f), dtype=np.float64)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    bad = lib.wmf_oracle_half_step(vp(Y), ctypes.c_int64(Y.shape[0]), ctypes.c_int(f), ctypes.c_int(int(bias)), vp(indptr),
                                   vp(indices), vp(values), ctypes.c_int64(n), ctypes.c_double(lam), vp(X))
    if bad:
        raise ArithmeticError(f"{bad} singular rows")
    return X
