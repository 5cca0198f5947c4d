"""The matrix-free iteration kernel (csrc/wmf_iter.hip, round 4) on every path it can take, at every geometry it has.

A row with 33 .. wmf_iter_dmax entries is solved by a truncated polynomial in its operator E = V_u^T D V_u when tr E bounds it
close enough to the identity: the Neumann series while tr E <= 0.8, the Chebyshev recurrence when the series contracts slowly
or tr E is larger (condition bound <= 4), and it is HANDED BACK to the elimination kernels otherwise.  tr E ~ w d f / m, so
the fixed side's size m and the weight scale steer the path: these tests build small matrices against a LARGE fixed side
(whitened rows of norm^2 ~ f / m) and scale the weights, read the path shares back (wmf_plan_iter_stats) to make sure the
path under test really ran, compare sampled rows with the float64 oracle (RecModel/wmf_model.py:231-239) at the gates of
the elimination kernels, and demand bit-identical rows from two runs."""
import numpy as np
import pytest
import torch

from conftest import record_error
from oracle import wmf_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(n_rows, m_fixed, k, bias, deg_lo, deg_hi, scale, seed, neg_frac=0.0):
    from recmodel_amd import WMF
    from recmodel_amd.engine import AlsEngine
    rng = np.random.default_rng(seed)
    deg = rng.integers(deg_lo, deg_hi + 1, n_rows)
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([np.sort(rng.choice(m_fixed, d, replace=False)) for d in deg]).astype(np.int64)
    w = (scale * 10 * np.log(1 + rng.integers(1, 8, indptr[-1]))).astype(np.float32)
    w[rng.random(w.size) < 0.02] = 0.0                              # stored zeros contribute p = 1 (wmf_model.py:232, :239)
    eng = AlsEngine(n_rows, m_fixed, k, bias, 0.1)
    eng.set_interactions(torch.from_numpy(indptr).cuda(), torch.from_numpy(indices).cuda(), torch.from_numpy(w).cuda())
    Y = WMF(num_items=m_fixed, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias, seed=seed).items
    if bias:                                                        # the fixed side's biases: small, or (neg_frac) large enough
        Y[:, 0] *= 0.5                                              # to push a share of the weights below zero
        if neg_frac:
            Y[:, 0] = np.where(rng.random(m_fixed) < neg_frac, 25.0 * scale, Y[:, 0])
    eng.set_factors("items", Y)
    return eng, indptr, indices, w, Y


def _check(eng, indptr, indices, w, Y, rows, gate_row, gate_fro, name):
    f, bias = eng.f, eng.bias
    Yd = Y.astype(np.float64)
    Gy = Yd.copy()
    if bias:
        Gy[:, 0] = 1.0
    G = Gy.T @ Gy + eng.gamma * np.eye(f)
    got = eng.get_factors("users").astype(np.float64)
    worst, num, den = 0.0, 0.0, 0.0
    for u in rows:
        lo, hi = indptr[u], indptr[u + 1]
        idx, ww = indices[lo:hi], w[lo:hi].astype(np.float64)
        U = Gy[idx]
        if bias:
            ww = ww - Yd[idx, 0]
        want = orc.solve_row(G, U, np.arange(hi - lo), ww)
        e, n_ = np.linalg.norm(got[u] - want), np.linalg.norm(want)
        worst = max(worst, e / n_)
        num += e * e
        den += n_ * n_
    fro = float(np.sqrt(num / den))
    record_error(name, worst_row=worst, fro=fro)
    assert worst <= gate_row and fro <= gate_fro, (name, worst, fro)


# (k, bias): the kernel's geometries -- 4 waves x 8 features with and without the split layout's border (k = 128: the LDS-DMA
# variant), pieces that end inside a row (k = 100), 8 waves x 12 / 16 / 20 features (k = 160, 256, 260)
GEOMETRIES = [(128, True), (128, False), (100, False), (160, True), (256, False), (260, False)]


@pytest.mark.parametrize("k,bias", GEOMETRIES)
def test_iteration_paths_neumann_chebyshev_and_hand_back(k, bias):
    m_fixed, n_rows = 400_000, 1500
    f = k + int(bias)
    dmax = 128 if k == 128 else (144 if k == 100 else (192 if k == 260 else 256))
    tau1 = 15.0 * 60 * f / m_fixed                                 # tr E of a 60-entry row at scale 1, roughly
    seen = set()
    for label, scale in (("neumann", 0.01 / tau1), ("chebyshev", 1.5 / tau1), ("hand_back", 40.0 / tau1)):
        # (the Neumann case covers every register slot of the geometry -- rows up to its longest --, the other two keep tr E in a
        # band: rows of 33 .. 80 entries)
        eng, indptr, indices, w, Y = _engine(n_rows, m_fixed, k, bias, 33, dmax if label == "neumann" else 80, scale, seed=100 + k)
        eng.half_step("users")
        eng.check_numerics()
        eng.iter_stats("users")                                     # (clear: the counters below are those of ONE half step)
        first = eng.get_factors("users").copy()
        eng.half_step("users")
        done, bounced, apps, cheb = (int(x) for x in eng.iter_stats("users"))
        assert np.array_equal(first, eng.get_factors("users")), f"{label}: two runs differ"
        assert done + bounced == n_rows, (label, done, bounced)
        if label == "neumann":
            assert bounced == 0 and cheb == 0 and apps <= 4 * done, (done, bounced, apps, cheb)
        elif label == "chebyshev":
            assert cheb > 0.5 * n_rows, (done, bounced, apps, cheb)
        else:
            assert bounced > 0.9 * n_rows, (done, bounced, apps, cheb)
        seen.add(label)
        rows = np.random.default_rng(k).choice(n_rows, 150, replace=False)
        gate = (1.5e-5, 7e-6) if f <= 144 else (5e-5, 1.7e-5)      # <= 3 x measured (profiles/r04_parity_errors.json: 4.8e-6 / 2.3e-6, 1.6e-5 / 5.4e-6)
        _check(eng, indptr, indices, w, Y, rows, *gate, name=f"iter_paths[k={k},bias={int(bias)},{label}]")
        del eng
        torch.cuda.empty_cache()
    assert seen == {"neumann", "chebyshev", "hand_back"}


def test_iteration_with_negative_weights():
    """A bias model whose fixed-side biases push a share of the weights below zero: the row's operator is indefinite, the
    spectrum bound becomes [1 - tau_minus, 1 + tau_plus]; rows stay on the iteration while tau_minus <= 1/2 and go back to the
    elimination (and from there to the pivoted kernel) beyond."""
    k, m_fixed, n_rows = 128, 400_000, 1200
    for neg_frac, scale in ((0.05, 0.3), (0.5, 30.0)):
        eng, indptr, indices, w, Y = _engine(n_rows, m_fixed, k, True, 40, 120, scale, seed=7, neg_frac=neg_frac)
        eng.half_step("users")
        eng.iter_stats("users")
        eng.half_step("users")
        eng.check_numerics()
        done, bounced, apps, cheb = (int(x) for x in eng.iter_stats("users"))
        if neg_frac < 0.1:
            assert done > 0.9 * n_rows, (done, bounced)
        else:
            assert bounced > 0.9 * n_rows, (done, bounced)
        rows = np.random.default_rng(3).choice(n_rows, 120, replace=False)
        _check(eng, indptr, indices, w, Y, rows, 1.7e-5, 1e-5, name=f"iter_negative_weights[{neg_frac}]")   # measured 5.4e-6 / 3.2e-6
        del eng
        torch.cuda.empty_cache()


def test_iteration_off_gives_the_same_rows():
    """Debug flag 268435456 sends every row to the elimination kernels: the two solvers must agree to float32 rounding on rows
    the iteration solves (cfg3-like shape, scaled down)."""
    from recmodel_amd import _lib
    eng, indptr, indices, w, Y = _engine(4000, 300_000, 128, True, 33, 128, 0.5, seed=11)
    eng.half_step("users")
    a = eng.get_factors("users").copy()
    lib = _lib.load()
    try:
        lib.wmf_debug_set_flags(268435456)
        eng.half_step("users")
    finally:
        lib.wmf_debug_set_flags(0)
    b = eng.get_factors("users")
    rel = np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)
    record_error("iter_vs_elimination", worst_row=float(rel.max()), median_row=float(np.median(rel)))
    assert rel.max() <= 4.2e-6, rel.max()                         # measured 1.4e-6


def test_rolled_coordinates_carry_the_bias_in_the_rows_own_bits(monkeypatch):
    """k = 128 with biases (include/wmf_hip.h, wmf_row_transform modes 3 / 4, wmf_solve_rows_ex): the whitened side is kept in
    coordinates rolled by one, so that its border feature is the constant 1 / L00, and the 32 bits of a row's bias replace the last
    mantissa bit of body positions 8 j, 8 j + 1 -- the iteration kernels then fetch nothing from the {border, bias} pairs.  Checked:
    the engine takes that path by default; every border value is the same number; the bits in the body ARE the bias, bit for bit;
    the rows agree with the plain split layout (WMF_ROLLED=0) to float32 rounding and with the float64 oracle at the usual gate."""
    eng, indptr, indices, w, Y = _engine(3000, 250_000, 128, True, 33, 120, 1.0, 777)
    assert eng.rolled and eng.white_mode == 3 and eng.solve_flags == 1
    eng.half_step("users")
    torch.cuda.synchronize()
    st = eng.iter_stats("users")
    assert int(st[0]) >= 2900, st                                   # the rows went through the iteration kernel
    body = eng.V["items"].cpu().numpy().reshape(-1, 128)[: Y.shape[0]]
    pairs = eng.bias_vec["items"].cpu().numpy().reshape(-1, 2)[: Y.shape[0]]
    assert np.unique(pairs[:, 0]).size == 1                         # the border feature: one number for every row
    np.testing.assert_array_equal(pairs[:, 1], Y[:, 0].astype(np.float32))
    bits = body.view(np.uint32) & 1
    rebuilt = np.zeros(Y.shape[0], dtype=np.uint32)
    for j in range(16):
        rebuilt |= (bits[:, 8 * j] << np.uint32(2 * j)) | (bits[:, 8 * j + 1] << np.uint32(2 * j + 1))
    np.testing.assert_array_equal(rebuilt.view(np.float32), pairs[:, 1])
    rolled = eng.factors["users"].cpu().numpy()[:3000, : eng.f].copy()
    _check(eng, indptr, indices, w, Y, np.arange(0, 3000, 23), 1.5e-5, 7e-6, "rolled[k=128,bias=1]")
    monkeypatch.setenv("WMF_ROLLED", "0")
    plain_eng, *_ = _engine(3000, 250_000, 128, True, 33, 120, 1.0, 777)
    assert not plain_eng.rolled
    plain_eng.half_step("users")
    torch.cuda.synchronize()
    plain = plain_eng.factors["users"].cpu().numpy()[:3000, : eng.f]
    diff = np.linalg.norm(rolled - plain, axis=1) / np.linalg.norm(plain, axis=1)
    record_error("rolled_vs_plain[k=128,bias=1]", worst_row=float(diff.max()))
    assert diff.max() <= 4.2e-6, diff.max()


@pytest.mark.parametrize("k,bias", [(64, False), (64, True), (100, False), (128, True)])
def test_float64_iteration_against_the_float64_oracle(k, bias):
    """The float64 form of the iteration (csrc/wmf_iter64.hip, inside wmf_half_step_f64 -- the reference's cores > 1 variants,
    wmf_model.py:242-309): rows of 1 .. 32 entries (one wave per row) and of 33 .. 256 / 192 / 144 entries (four waves)
    against a LARGE fixed side, so that tr E is small and the series runs; 1e-10 of oracle.solve_row like every float64 test, and
    the same rows with the iteration switched off (debug flag 268435456: blocked Cholesky / low-rank kernels) agree to 1e-12, and so
    do the matrix-core forms of the Gramian and the row transform with their VALU forms (debug flag 536870912)."""
    from recmodel_amd import _lib
    from recmodel_amd.engine import HipKernels
    f = k + int(bias)
    rng = np.random.default_rng(500 + f)
    m_fixed, n_rows = 300_000, 1200
    dmax = {64: 256, 65: 192, 100: 144, 129: 144}[f]
    deg = np.concatenate([rng.integers(1, 33, n_rows // 2), rng.integers(33, dmax + 1, n_rows - n_rows // 2)])
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([np.sort(rng.choice(m_fixed, d, replace=False)) for d in deg]).astype(np.int32)
    w = 10 * np.log(1 + rng.integers(1, 8, indptr[-1])).astype(np.float64)
    w[rng.random(w.size) < 0.02] = 0.0
    Y = rng.random((m_fixed, f))
    if bias:
        Y[:, 0] *= 0.5
    K = HipKernels()
    dev = torch.device("cuda:0")
    Yd = torch.from_numpy(Y).to(dev)
    ip, ix, wd = torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev), torch.from_numpy(w).to(dev)
    ws = torch.empty(K.half_step_f64_workspace_bytes(f, m_fixed, n_rows), dtype=torch.uint8, device=dev)
    fail = torch.zeros(4, dtype=torch.int32, device=dev)
    outs = []
    lib = _lib.load()
    for flags in (0, 268435456, 536870912):
        out = torch.empty(n_rows, f, dtype=torch.float64, device=dev)
        try:
            lib.wmf_debug_set_flags(flags)
            K.half_step_f64(Yd, m_fixed, f, bias, ip, ix, wd, n_rows, 0.1, out, ws, fail)
            torch.cuda.synchronize()
        finally:
            lib.wmf_debug_set_flags(0)
        assert int(fail[0]) == 0
        outs.append(out.cpu().numpy())
    Gy = Y.copy()
    if bias:
        Gy[:, 0] = 1.0
    G = Gy.T @ Gy + 0.1 * np.eye(f)
    worst = 0.0
    for u in rng.choice(n_rows, 160, replace=False):
        lo, hi = indptr[u], indptr[u + 1]
        idx, ww = indices[lo:hi], w[lo:hi].copy()
        if bias:
            ww = ww - Y[idx, 0]
        want = orc.solve_row(G, Gy[idx], np.arange(hi - lo), ww)
        worst = max(worst, np.linalg.norm(outs[0][u] - want) / np.linalg.norm(want))
    both = np.linalg.norm(outs[0] - outs[1], axis=1) / np.linalg.norm(outs[1], axis=1)
    dense = np.linalg.norm(outs[0] - outs[2], axis=1) / np.linalg.norm(outs[2], axis=1)
    record_error(f"iter64[k={k},bias={int(bias)}]", worst_row=float(worst), vs_direct_kernels=float(both.max()),
                 vs_valu_dense_forms=float(dense.max()))
    assert worst <= 1e-10 and both.max() <= 1e-12 and dense.max() <= 1e-12, (worst, both.max(), dense.max())
