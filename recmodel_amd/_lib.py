"""ctypes binding of libwmf_hip.so (include/wmf_hip.h).

The library is the product: if it cannot be loaded, or no HIP device is present when a compute
entry point is called, this module raises -- there is no CPU fallback.

torch is imported before the library on purpose: the PyTorch-ROCm wheel ships its own
libamdhip64.so (same SONAME as /opt/rocm's).  Loading torch first makes the dynamic loader bind
libwmf_hip.so to that already-loaded runtime, so torch tensors, torch streams and our kernels live
in ONE HIP runtime inside the process.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WMF_HIP_LIB", os.path.join(_HERE, "libwmf_hip.so"))   # override: kernel tuning experiments

WMF_OK, WMF_EINVAL, WMF_EHIP, WMF_ENOMEM, WMF_ENUMERIC = 0, -1, -2, -3, -4

c_int, c_i64, c_dbl, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/wmf_hip.h one to one
SIGNATURES = {
    "wmf_last_error": (ctypes.c_char_p, []),
    "wmf_version": (c_int, []),
    "wmf_ld_for": (c_int, [c_int]),
    "wmf_recompute_factors_host": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_dbl, c_vp]),
    "wmf_gram_workspace_bytes": (c_i64, [c_int]),
    "wmf_gram": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "wmf_factorize": (c_int, [c_vp, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wmf_row_transform": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "wmf_whitened_row_floats": (c_int, [c_int, c_int, c_int]),
    "wmf_plan_create": (c_int, [c_vp, c_i64, c_int, c_int, ctypes.POINTER(c_vp)]),
    "wmf_plan_destroy": (None, [c_vp]),
    "wmf_plan_stats": (c_int, [c_vp, c_vp]),
    "wmf_plan_iter_stats": (c_int, [c_vp, c_vp]),
    "wmf_solve_rows": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp]),
    "wmf_solve_rows_ex": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_int, c_vp]),
    "wmf_rolled_layout_supported": (c_int, [c_int, c_int]),
    "wmf_eval_workspace_bytes": (c_i64, []),
    "wmf_eval_sqerr": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "wmf_predict_pairs": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "wmf_partial_row_floats": (c_i64, [c_int]),
    "wmf_accumulate_rows": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_int, c_vp, c_int, c_int, c_vp, c_vp]),
    "wmf_eliminate_rows": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "wmf_rank_workspace_bytes": (c_i64, [c_i64]),
    "wmf_rank_topn": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wmf_rank_batch_workspace_bytes": (c_i64, [c_i64, c_i64]),
    "wmf_rank_topn_batch": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wmf_hit_counts": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp]),
    "wmf_spmm_rows": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "wmf_gather_rows": (c_int, [c_vp, c_int, c_vp, c_i64, c_vp, c_vp]),
    "wmf_coo_to_csr_workspace_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "wmf_coo_to_csr": (c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "wmf_confidence_transform": (c_int, [c_vp, c_i64, c_dbl, c_dbl, c_int, c_vp]),
    "wmf_confidence_transform_f64": (c_int, [c_vp, c_i64, c_dbl, c_dbl, c_int, c_vp]),
    "wmf_half_step_f64_workspace_bytes": (c_i64, [c_int, c_i64, c_i64]),
    "wmf_half_step_f64": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_dbl, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "wmf_recompute_factors_f64_host": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_i64, c_dbl, c_vp]),
    "wmf_profile_enable": (c_int, [c_int]),
    "wmf_profile_set_tag": (c_int, [c_int]),
    "wmf_profile_collect": (c_int, []),
    "wmf_profile_entry": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "wmf_profile_reset": (c_int, []),
    "wmf_debug_set_flags": (c_int, [c_int]),
}


class WmfLibraryError(RuntimeError):
    """libwmf_hip.so is missing or a HIP call failed."""


class WmfNumericError(ArithmeticError):
    """The Gramian was not positive definite or a row system was singular."""


_lib = None


def load():
    """Load libwmf_hip.so and declare every symbol of the header.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WmfLibraryError(
            f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C recmodel_amd/csrc`.  There is no CPU fallback for the WMF hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header and library out of sync
        fn.restype, fn.argtypes = res, args
    # kernel-SELECTION switches for experiments (include/wmf_hip.h, wmf_debug_set_flags): every selection computes the same
    # results; e.g. WMF_DEBUG_FLAGS=268435456 runs the whole parity suite without the matrix-free iteration kernel
    flags = os.environ.get("WMF_DEBUG_FLAGS")
    if flags:
        lib.wmf_debug_set_flags(int(flags, 0))
    _lib = lib
    return lib


def profile_table(lib=None):
    """[(kernel symbol, tag, total ms, launches, min ms, max ms)] of the launches recorded since the last reset
    (wmf_profile_collect / wmf_profile_entry)."""
    lib = lib or load()
    out = []
    name = ctypes.create_string_buffer(200)
    tag, n = c_int(), c_i64()
    ms, lo, hi = c_dbl(), c_dbl(), c_dbl()
    for i in range(lib.wmf_profile_collect()):
        check(lib.wmf_profile_entry(i, name, 200, ctypes.byref(tag), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(lo),
                                    ctypes.byref(hi)))
        out.append((name.value.decode(), tag.value, ms.value, n.value, lo.value, hi.value))
    return out


def check(rc):
    """Translate a WMF_E* return code into the exception type the reference raises for the same misuse."""
    if rc == WMF_OK:
        return
    msg = load().wmf_last_error().decode("utf-8", "replace")
    if rc == WMF_EINVAL:
        raise ValueError(msg)
    if rc == WMF_ENOMEM:
        raise MemoryError(msg)
    if rc == WMF_ENUMERIC:
        raise WmfNumericError(msg)
    raise WmfLibraryError(msg)


def require_gpu():
    if not torch.cuda.is_available():
        raise WmfLibraryError("no HIP device visible: the WMF hot path runs on MI355X only (no CPU fallback)")
