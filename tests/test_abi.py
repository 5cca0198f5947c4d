"""CPU-only checks of the C ABI: the library builds, loads, and exports every symbol that
include/wmf_hip.h declares; argument validation that needs no GPU; ctypes table in sync."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "wmf_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build()
    from recmodel_amd import _lib
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wmf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in wmf_hip.h but not exported by libwmf_hip.so"


def test_ctypes_table_matches_header(lib):
    from recmodel_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "wmf_hip.h"\nint main(void){ return WMF_OK + (int)sizeof(int64_t) - 8; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                           "-o", str(tmp_path / "t.o")])


def test_header_cites_reference_lines():
    text = open(HEADER).read()
    for cite in ("wmf_model.py:213-240", "wmf_model.py:311-351", "base_model.py:150-179", "wmf_model.py:191-211"):
        assert cite in text


def test_pure_host_entry_points(lib):
    assert lib.wmf_version() >= 100
    assert [lib.wmf_ld_for(f) for f in (1, 16, 64, 65, 129, 257)] == [4, 16, 64, 68, 132, 260]
    assert lib.wmf_gram_workspace_bytes(64) > 0 and lib.wmf_gram_workspace_bytes(0) == 0
    assert lib.wmf_eval_workspace_bytes() > 0
    assert lib.wmf_profile_reset() == 0 and lib.wmf_profile_collect() == 0       # nothing recorded: an empty table
    assert lib.wmf_profile_entry(0, None, 0, None, None, None, None, None) == -1  # WMF_EINVAL, no entry 0
    assert lib.wmf_debug_set_flags(2) == -1                                      # ablation switches: -DWMF_LAB builds only
    assert lib.wmf_debug_set_flags(0) == 0
    # the split layout of the whitened fixed side (k = 16 m with biases, m + 1 not a multiple of 4): the same predicate in
    # the library and in the CPU stand-in the host-logic tests run on
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_kernels import NumpyKernels
    fk = NumpyKernels()
    for f in list(range(1, 150)) + [161, 193, 257]:
        ld = lib.wmf_ld_for(f)
        for bias in (0, 1):
            assert lib.wmf_whitened_row_floats(f, ld, bias) == fk.whitened_row_floats(f, ld, bool(bias)), (f, bias)
    assert [lib.wmf_whitened_row_floats(f, lib.wmf_ld_for(f), 1) for f in (17, 33, 49, 65, 113, 129, 145)] == [16, 32, 52, 64, 116, 128, 148]
    assert lib.wmf_whitened_row_floats(129, 132, 0) == 132


def test_argument_validation_without_gpu(lib):
    from recmodel_amd import _lib
    dummy = ctypes.c_void_p(16)
    # ld not a multiple of 4 / smaller than f -> WMF_EINVAL -> ValueError, before any HIP call
    with pytest.raises(ValueError):
        _lib.check(lib.wmf_gram(dummy, 10, 64, 63, 0, dummy, dummy, None))
    with pytest.raises(ValueError):
        _lib.check(lib.wmf_row_transform(dummy, 10, 300, 300, dummy, 0, dummy, None, None))
    with pytest.raises(ValueError):   # the reference's predict() length rule (wmf_model.py:200-203)
        _lib.check(lib.wmf_predict_pairs(dummy, dummy, 16, 16, 0, dummy, 3, dummy, 2, dummy, None))
    with pytest.raises(ValueError):
        _lib.check(lib.wmf_confidence_transform(dummy, 5, 10.0, 1.0, 7, None))
    bad_ptr = np.array([0, 2, 1], dtype=np.int64)   # not monotone
    plan = ctypes.c_void_p()
    with pytest.raises(ValueError):
        _lib.check(lib.wmf_plan_create(bad_ptr.ctypes.data_as(ctypes.c_void_p), 2, 16, 0, ctypes.byref(plan)))


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import scipy.sparse as sp
    from recmodel_amd import WMF, _lib
    m = WMF(num_items=5, num_users=4, dim=2, gamma=0.1, weighted=True)
    c = sp.random(4, 5, density=0.5, format="csr", random_state=0)
    with pytest.raises(_lib.WmfLibraryError):
        m.train(utility_mat=c, iterations=1, eval_mat=c, count_mat=c)
    with pytest.raises(_lib.WmfLibraryError):
        m.recompute_factors(m.items, c, 0.1)


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "recmodel_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("test infrastructure", ""), f"{fn} mentions the oracle"


def test_lds_dma_kernel_leaves_in_flight_registers_alone():
    """wmf_directl.hip reads its LDS ring with inline-asm ds_reads whose destination registers hipcc believes written at
    once; nothing may touch them before the inline-asm wait that retires them (tools/check_inflight_regs.py parses the
    gfx950 assembly of the file -- no GPU needed)."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tools", "check_inflight_regs.py")
    res = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "0 violations" in res.stdout
