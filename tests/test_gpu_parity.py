"""Parity of the HIP path against (a) golden vectors captured from the reference WMF class and
(b) the CPU oracle on seeded inputs.  Everything here calls through the C ABI of libwmf_hip.so.

Stated fp32 tolerance (DESIGN.md "Numerics"): the reference computes each row system in float64
when the count matrix is float64 and stores float32; the HIP path computes in float32 on
whitened factors.  Its error is bounded by ~cond(A_u) * eps_f32:
  * one half step from identical inputs:  relative Frobenius error <= 5e-5, worst row <= 5e-4
    (the worst case is the bias model at its uniform(0,1) initialisation, cond(G) ~ 1e4);
  * after T >= 2 iterations: relative Frobenius error <= 1e-3, |dMSE|/MSE <= 1e-4.
"""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import csr_from, load_golden, record_error
from oracle import wmf_oracle as orc

pytestmark = pytest.mark.gpu

HALF_FRO, HALF_ROW = 5e-5, 5e-4
WIDE_FRO, WIDE_ROW = 1.5e-4, 1e-3          # f > 144 (k = 256): conditioning grows with f; DESIGN.md section 2 numerics
TRAIN_FRO, TRAIN_MSE = 1e-3, 1e-4
# (worst row, relative Frobenius) of test_half_step_vs_oracle_all_degree_classes per (k, bias): 3 x the values measured on
# MI355X (profiles/r03_parity_errors.json) -- a regression of 3 x fails even where the stated tolerance above would still pass
DEGREE_GATES = {
    (16, False): (9.7e-06, 6.1e-06), (16, True): (3.6e-05, 1.3e-05), (32, True): (1.1e-04, 3.5e-05), (33, True): (1.3e-04, 3.7e-05),
    (48, True): (1.5e-04, 4.2e-05), (50, False): (2e-05, 1.4e-05), (64, False): (2.2e-05, 1.6e-05), (64, True): (2.9e-04, 6.4e-05),
    (96, True): (2.7e-04, 7.3e-05), (112, True): (3.5e-04, 9.8e-05), (128, False): (3.8e-05, 2.7e-05), (128, True): (4.9e-04, 1.2e-04),
    (150, False): (1.4e-04, 3e-05), (176, True): (4.2e-04, 1.5e-04), (200, True): (7e-04, 2.2e-04), (240, True): (8.4e-04, 2.8e-04),
    (256, False): (3.2e-04, 6.6e-05), (256, True): (7.7e-04, 2.7e-04),
}


def fro(a, b):
    return np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)


def worst_row(a, b):
    num = np.linalg.norm(a.astype(np.float64) - b, axis=1)
    den = np.linalg.norm(b, axis=1)
    ok = den > 0
    return (num[ok] / den[ok]).max() if ok.any() else 0.0, (num[~ok].max() if (~ok).any() else 0.0)


@pytest.fixture(scope="module")
def WMF():
    from recmodel_amd import WMF as cls, _lib
    _lib.load()
    return cls


# ------------------------------------------------------------------ golden vectors of the reference
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("cdt", ["float32", "float64"])
def test_half_steps_match_reference_golden(WMF, bias, cdt):
    g = load_golden(f"half_bias{int(bias)}_{cdt}.npz")
    C, CT = csr_from(g, "C"), csr_from(g, "CT")
    m = WMF(num_items=C.shape[1], num_users=C.shape[0], dim=16, gamma=0.1, weighted=True, bias=bias)
    step = m.recompute_factors_bias if bias else m.recompute_factors
    for Y, mat, want in ((g["items0"], C, g["users1"]), (g["users1"], CT, g["items1"]), (g["items1"], C, g["users2"])):
        got = step(Y, mat, 0.1)
        assert got.dtype == np.float32 and got.shape == want.shape
        assert fro(got, want) <= HALF_FRO
        rel, zero_abs = worst_row(got, want)
        assert rel <= HALF_ROW and zero_abs == 0.0      # rows without stored entries are exactly zero
    assert np.all(step(g["items0"], C, 0.1)[3] == 0)    # the fixture's empty user row


@pytest.mark.parametrize("tag,bias", [("run", False), ("stop", False), ("bias", True)])
def test_train_matches_reference_golden(WMF, tag, bias, capsys):
    g = load_golden(f"train_{tag}.npz")
    counts, util = csr_from(g, "counts"), csr_from(g, "util")
    m = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=int(g["dim"]), gamma=0.1, weighted=True, bias=bias)
    hist = []
    inner = m.eval_prec
    last = m.train(utility_mat=util, iterations=int(g["iterations"]), eval_mat=util, count_mat=counts, cores=1,
                   stopping_rounds=int(g["stopping_rounds"]), min_improvement=float(g["min_improvement"]))
    out = capsys.readouterr().out
    assert last == int(g["last_iter"])                               # early-stop control flow (wmf_model.py:164-180)
    assert ("took" in out) == bias                                   # the bias path prints per-iteration time (:153)
    assert m.users.dtype == np.float32 and m.items.dtype == np.float32
    assert fro(m.users, g["users"]) <= TRAIN_FRO and fro(m.items, g["items"]) <= TRAIN_FRO
    mse = inner(util)
    assert abs(mse - float(g["mse_final"])) <= TRAIN_MSE * float(g["mse_final"])
    assert abs(m.eval_prec(util, "rmse") - float(g["rmse_final"])) <= TRAIN_MSE * float(g["rmse_final"])
    assert abs(m.eval_prec(util, "mae") - float(g["mae_final"])) <= TRAIN_MSE * float(g["mae_final"])
    with pytest.raises(ValueError):
        m.eval_prec(util, "auc")


def test_predict_rank_on_reference_factors(WMF):
    for tag, bias in (("run", False), ("bias", True)):
        g = load_golden(f"train_{tag}.npz")
        util = csr_from(g, "util")
        m = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=int(g["dim"]), gamma=0.1, weighted=True, bias=bias)
        m.users, m.items = g["users"], g["items"]                    # the reference's own final factors
        np.testing.assert_allclose(m.predict(g["pred_users"], g["pred_items"]), g["pred"], rtol=2e-6, atol=2e-6)
        cand = g["rank_cand"]
        np.testing.assert_allclose(m.predict(np.array([7]), cand), g["pred_one_user"], rtol=2e-6, atol=2e-6)
        with pytest.raises(ValueError):      # the reference's length check only lets ONE USER broadcast (wmf_model.py:202)
            m.predict([3, 4, 5], [9])
        np.testing.assert_allclose(m.predict([3], [9, 10, 11]), orc.predict(m.users, m.items, [3, 3, 3], [9, 10, 11], bias), rtol=2e-6, atol=2e-6)
        # rank: tie order is implementation defined -> compare the scores of the returned ids
        sc = dict(zip(cand.tolist(), g["pred_one_user"].tolist()))
        for topn, key in ((5, "rank_top5"), (40, "rank_top40"), (None, "rank_all")):
            got = m.rank(cand, 7, topn=topn)
            assert len(got) == len(g[key])
            np.testing.assert_allclose([sc[i] for i in got], [sc[i] for i in g[key]], rtol=1e-5, atol=1e-6)
        lst = m.rank(cand, [1, 2], topn=3)
        assert isinstance(lst, list) and len(lst) == 2
        with pytest.raises(ValueError):
            m.predict(np.array([0, 1]), np.array([0, 1, 2]))
        np.testing.assert_allclose(m.eval_prec(util), float(g["mse_final"]), rtol=1e-5)


@pytest.mark.parametrize("bias", [False, True])
def test_eval_topn_matches_reference_golden(WMF, bias):
    """Sampled Recall@N on the device (wmf_hit_counts: every test entry ranked among its user's random
    candidates in one launch) against the reference's seeded eval_topn on the reference's own factors."""
    g = load_golden(f"eval_topn_bias{int(bias)}.npz")
    test = csr_from(g, "test")
    m = WMF(num_items=test.shape[1], num_users=test.shape[0], dim=8, gamma=0.1, weighted=True, bias=bias)
    m.users, m.items = g["users"], g["items"]
    res = m.eval_topn(test_mat=test.copy(), topn=g["topn"], rand_sampled=int(g["rand_sampled"]), random_state=int(g["random_state"]))
    got = np.array([res[f"Recall@{n}"] for n in g["topn"]], dtype=np.float64)
    np.testing.assert_array_equal(got, g["recall"])
    # the entry-by-entry route through rank() (what a model without the device hook takes) gives the same hits
    np.random.seed(int(g["random_state"]))
    hits = np.zeros(len(g["topn"]), dtype="float32")
    from recmodel_amd.base_model import iter_rows_two_matrices
    for elem in iter_rows_two_matrices(test, test):
        hits += m.compute_hit(elem, rand_sampled=int(g["rand_sampled"]), topn=g["topn"])
    np.testing.assert_array_equal((hits / int(g["n_test"])).astype(np.float64), g["recall"])
    with pytest.raises(ValueError):
        m.eval_topn(test_mat=test, topn=[1, 5])


def test_rank_full_catalogue_on_device():
    """rank over a 200 000-item catalogue: device scores + device sort vs NumPy on the same factors."""
    from recmodel_amd import WMF
    rng = np.random.default_rng(5)
    n_items, n_users, k = 200_000, 64, 32
    m = WMF(num_items=n_items, num_users=n_users, dim=k, gamma=0.1, weighted=True, bias=True)
    m.users = rng.standard_normal((n_users, k + 1)).astype(np.float32)
    m.items = rng.standard_normal((n_items, k + 1)).astype(np.float32)
    cand = rng.permutation(n_items).astype(np.int64)
    ref = orc.predict(m.users.astype(np.float64), m.items.astype(np.float64), [3], cand, True)
    for topn in (10, 150_000, None):
        got = m.rank(cand, 3, topn=topn)
        want = orc.rank(m.users.astype(np.float64), m.items.astype(np.float64), cand, [3], topn=topn, bias=True)
        assert len(got) == len(want) and len(set(got.tolist())) == len(got)
        sc = dict(zip(cand.tolist(), ref.tolist()))
        gs = np.array([sc[i] for i in got])
        np.testing.assert_allclose(gs, [sc[i] for i in want], rtol=0, atol=2e-5)       # same scores position by position
        assert (np.diff(gs) <= 2e-5).all()                                                  # best first
    # a list of users: one batched device call (MFMA score tiles + segmented sort) instead of the reference's loop
    sub = cand[:5000]
    Uf, If = m.users.astype(np.float64), m.items.astype(np.float64)
    who = [3, 0, 63, 17, 3] + list(range(20, 45))
    many = m.rank(sub, who, topn=25)
    assert isinstance(many, list) and len(many) == len(who)
    for usr, got in zip(who, many):
        sc_u = orc.predict(Uf, If, [usr], sub, True)
        lut = dict(zip(sub.tolist(), sc_u.tolist()))
        np.testing.assert_allclose([lut[i] for i in got], np.sort(sc_u)[::-1][:25], rtol=0, atol=3e-5)
    # ... and agrees with the one-user entry point
    sc17 = dict(zip(sub.tolist(), orc.predict(Uf, If, [17], sub, True).tolist()))
    np.testing.assert_allclose([sc17[i] for i in m.rank(sub, 17, topn=25)], [sc17[i] for i in many[3]], rtol=0, atol=3e-5)
    assert m.rank(cand[:7], [], topn=3) == []
    assert len(m.rank(cand[:5], 3, topn=50)) == 5                                            # topn beyond the list: all of it
    with pytest.raises(IndexError):
        m.rank(np.array([0, n_items]), 3, topn=1)


def test_rank_select_with_ties_and_extremes():
    """The top-n select of wmf_rank_topn where its histogram cannot separate the candidates: every score equal (the list
    that is sorted is the whole list; ties come out in candidate order), two score levels with the cut inside the lower
    one, infinities, and topn = 1 / n_cand."""
    from recmodel_amd import WMF
    n_items, k = 30_000, 8
    m = WMF(num_items=n_items, num_users=2, dim=k, gamma=0.1, weighted=True, bias=False)
    m.users = np.zeros((2, k), dtype=np.float32); m.users[0, 0] = 1.0
    m.items = np.zeros((n_items, k), dtype=np.float32)
    cand = np.arange(n_items)[::-1].copy()                               # candidate order != item order
    got = m.rank(cand, 0, topn=7)                                        # all scores 0
    np.testing.assert_array_equal(got, cand[:7])
    m.items[::3, 0] = 2.0                                                # 10 000 items score 2, the others 0
    m.items = m.items.copy()
    hi = [i for i in cand if i % 3 == 0]
    lo = [i for i in cand if i % 3 != 0]
    np.testing.assert_array_equal(m.rank(cand, 0, topn=10_003), np.array(hi + lo[:3]))
    np.testing.assert_array_equal(m.rank(cand, 0, topn=1), np.array(hi[:1]))
    np.testing.assert_array_equal(m.rank(cand, 0, topn=n_items), np.array(hi + lo))
    m.items[5, 0], m.items[6, 0] = np.inf, -np.inf
    m.items = m.items.copy()
    r = m.rank(cand, 0, topn=n_items)
    assert r[0] == 5 and r[-1] == 6


def test_unweighted_branch_dtype_follows_the_utility_matrix(WMF):
    """wmf_model.py:85 / :88 are dense-times-sparse products: NumPy's result type of the model dtype (float32) and the utility
    matrix's -- float64 factors for SciPy's default float64 matrices (the golden), float32 ones for a float32 matrix (found by
    tests/scale/fuzz_train.py in round 3: the class returned float64 there)."""
    rng = np.random.default_rng(3)
    util = sp.random(90, 70, density=0.2, format="csr", random_state=5, data_rvs=lambda s: rng.integers(1, 5, s).astype(np.float64))
    for dt in (np.float32, np.float64, np.int64):
        u = util.astype(dt)
        m = WMF(num_items=70, num_users=90, dim=8, gamma=0.1, weighted=False, bias=False)
        m.train(utility_mat=u, iterations=2, eval_mat=u.astype(np.float32))
        _, _, wu, wi = orc.train(70, 90, 8, 0.1, u, 2, u.astype(np.float32), weighted=False)
        assert m.users.dtype == wu.dtype and m.items.dtype == wi.dtype, (dt, m.users.dtype, wu.dtype)
        assert fro(m.users, wu) <= TRAIN_FRO and fro(m.items, wi) <= TRAIN_FRO


def test_unweighted_branch_matches_reference_golden(WMF):
    g = load_golden("train_unweighted.npz")
    util = csr_from(g, "util")
    m = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=10, gamma=0.5, seed=1993)
    last = m.train(utility_mat=util, iterations=2, eval_mat=util, stopping_rounds=5)
    assert last == int(g["last_iter"]) and m.users.dtype == np.float64
    assert fro(m.users, g["users"]) <= 1e-4 and fro(m.items, g["items"]) <= 1e-4
    assert abs(m.eval_prec(util) - float(g["mse_final"])) <= 1e-4 * float(g["mse_final"])


# ------------------------------------------------------------------ oracle on seeded inputs
def as_f64(mat):
    """float64 copy that keeps the stored structure (scipy's astype() merges duplicate entries)."""
    return sp.csr_matrix((mat.data.astype(np.float64), mat.indices.copy(), mat.indptr.copy()), shape=mat.shape)


def ragged_matrix(n, m, seed, dtype=np.float32):
    """Rows of degree 0,1,16,17,32,33,100,max plus a Poisson body; one stored zero; one duplicate."""
    from recmodel_amd import synth
    rng = np.random.default_rng(seed)
    ip, idx, val = synth.make_counts(n, m, 11, seed)
    C = synth.to_scipy(ip, idx, val, (n, m)).tolil()
    for r, d in {0: 0, 1: 1, 2: 16, 3: 17, 4: 32, 5: 33, 6: 100, 7: m, 8: 2, 9: 31}.items():
        cols = np.sort(rng.choice(m, d, replace=False))
        C.rows[r] = list(cols)
        C.data[r] = list((2 + rng.integers(0, 5, d)).astype(dtype))
    C = C.tocsr().astype(dtype)
    C.data = (10 * np.log(1 + C.data)).astype(dtype)
    C.data[C.indptr[8]] = 0.0                                         # stored zero still contributes (w+1)*y
    # non-canonical CSR: append a duplicate of row 9's first entry at the end of row 9
    lo, hi = C.indptr[9], C.indptr[10]
    indices = np.concatenate([C.indices[:hi], C.indices[lo:lo + 1], C.indices[hi:]])
    data = np.concatenate([C.data[:hi], C.data[lo:lo + 1], C.data[hi:]])
    indptr = C.indptr.copy(); indptr[10:] += 1
    D = sp.csr_matrix((data, indices, indptr), shape=C.shape)
    assert D.nnz == C.nnz + 1
    return D


@pytest.mark.parametrize("k,bias", [(16, False), (64, False), (64, True), (128, False), (128, True), (50, False), (33, True),
                                    (256, False), (256, True), (150, False), (200, True),
                                    # f = 16 m + 1: the heavy-row kernel treats the last column as a border of an m-block system
                                    # (m = 1, 2, 6), except when m + 1 is a multiple of 4 (m = 3, 7: plain m + 1 blocks)
                                    (16, True), (32, True), (96, True), (48, True), (112, True),
                                    # the same border in the four-waves-per-row kernel (144 < f <= 257); 240: odd block count
                                    (176, True), (240, True)])
def test_half_step_vs_oracle_all_degree_classes(WMF, k, bias):
    n, m_items = (2500, 600) if k <= 128 else (900, 500)     # the NumPy oracle is O(f^3) per row
    C = ragged_matrix(n, m_items, seed=k + bias)
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    # one oracle iteration first so the factors are "trained-like" rather than uniform noise
    step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
    step_g = model.recompute_factors_bias if bias else model.recompute_factors
    C64, CT = as_f64(C), C.T.tocsr()
    for Y, mat in ((model.items, C), (step_o(model.items, C64, 0.1), CT)):
        want = step_o(Y, as_f64(mat), 0.1, out_dtype="float64")
        got = step_g(Y, mat, 0.1)
        rel, zero_abs = worst_row(got, want)
        tol_fro, tol_row = (HALF_FRO, HALF_ROW) if k + bias <= 144 else (WIDE_FRO, WIDE_ROW)
        tol_row, tol_fro = min(tol_row, DEGREE_GATES[(k, bias)][0]), min(tol_fro, DEGREE_GATES[(k, bias)][1])
        record_error(f"degree_classes[k={k},bias={int(bias)}] {'users' if mat is C else 'items'}", worst_row=rel, fro=fro(got, want))
        assert fro(got, want) <= tol_fro, (k, bias, fro(got, want))
        assert rel <= tol_row and zero_abs == 0.0, (k, bias, rel, zero_abs)
        assert not np.isnan(got).any()


@pytest.mark.parametrize("k,bias,neg", [(16, False, False), (64, False, False), (128, True, False), (130, False, False),
                                        (64, True, True)])
def test_short_rows_share_a_wave(WMF, k, bias, neg):
    """Rows with at most 8 stored entries are solved two per wave (solve_pair_kernel): an odd number of such rows,
    empty rows next to full ones, degrees 0..8 in every pairing, and -- with large fixed-side biases -- pairs in which
    one row's weights go negative, so that BOTH rows are bounced to the pivoted kernel."""
    from recmodel_amd import synth
    n, m_items = 1001, 300
    rng = np.random.default_rng(k + 7)
    ip, idx, val = synth.make_counts(n, m_items, 3, 4242 + k)
    C = synth.to_scipy(ip, idx, val, (n, m_items)).tolil()
    for r in range(0, 90):                                            # every (dA, dB) combination next to each other
        d = (r // 10, r % 10)[r % 2] % 9
        C.rows[r] = list(np.sort(rng.choice(m_items, d, replace=False)))
        C.data[r] = list((1 + rng.integers(0, 5, d)).astype(np.float32))
    if sum(len(r) <= 8 for r in C.rows) % 2 == 0:                     # make the number of short rows odd
        C.rows[100] = list(range(12))
        C.data[100] = [2.0] * 12
    C = C.tocsr().astype(np.float32)
    C.data = (10 * np.log(1 + C.data)).astype(np.float32)
    deg = np.diff(C.indptr)
    assert (deg <= 8).sum() % 2 == 1 and (deg == 0).sum() > 0
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
    step_g = model.recompute_factors_bias if bias else model.recompute_factors
    Y = step_o(step_o(model.items, as_f64(C), 0.1), as_f64(C.T.tocsr()), 0.1)       # trained-like item factors
    if neg:
        Y = Y.copy()
        Y[:, 0] = np.where(rng.random(m_items) < 0.15, 30.0, Y[:, 0])             # some weights w - bias go negative
    want = step_o(Y, as_f64(C), 0.1, out_dtype="float64")
    got = step_g(Y, C, 0.1)
    assert not np.isnan(got).any()
    if neg:
        ok = np.linalg.norm(want, axis=1) < 1e3
        assert ok.mean() > 0.9 and fro(got[ok], want[ok]) <= 1e-3
    else:
        rel, zero_abs = worst_row(got, want)
        assert fro(got, want) <= HALF_FRO and rel <= HALF_ROW and zero_abs == 0.0, (fro(got, want), rel, zero_abs)
        assert np.all(got[deg == 0] == 0)


@pytest.mark.parametrize("bias", [False, True])
def test_many_heavy_rows_per_wave_at_k128(WMF, bias):
    """More heavy rows than resident waves (8000 against 3072), 33 .. 400 entries each, k = 128: every wave of the LDS-DMA
    kernel walks several rows -- next row's metadata requested during the last group, its first rows during the elimination,
    the metadata buffers rotating across rows -- and must give what the ORACLE gives for every one of the 8000 rows (float64
    restatement of wmf_model.py:220-239 / :337-350), what the register-ring f32 kernel (debug flag 4096) gives, row by row,
    and satisfy the rows' own normal equations in float64."""
    from recmodel_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11 + bias)
    n, m_items, k = 8000, 3000, 128
    deg = rng.integers(33, 401, n)
    deg[:50] = rng.integers(33, 66, 50)
    rows = np.repeat(np.arange(n), deg)
    cols = np.concatenate([np.sort(rng.choice(m_items, d, replace=False)) for d in deg])
    vals = (10 * np.log(1 + rng.integers(1, 6, rows.size))).astype(np.float32)
    C = sp.csr_matrix((vals, (rows, cols)), shape=(n, m_items))
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    Y = model.items.copy()
    if bias:
        Y[:, 0] *= 0.5                                                 # weights stay positive: the SPD kernels are used
    step = model.recompute_factors_bias if bias else model.recompute_factors
    try:
        lib.wmf_debug_set_flags(4096)
        ref = step(Y, C, 0.1).astype(np.float64)
    finally:
        lib.wmf_debug_set_flags(0)
    got = step(Y, C, 0.1).astype(np.float64)
    want = (orc.recompute_factors_bias if bias else orc.recompute_factors)(Y, C.astype(np.float64), 0.1, out_dtype="float64")
    rel_o, zero_abs = worst_row(got, want)
    assert fro(got, want) <= HALF_FRO and rel_o <= HALF_ROW and zero_abs == 0.0, (fro(got, want), rel_o)
    try:                                                               # the one-wave-per-SIMD variant (16-entry groups)
        lib.wmf_debug_set_flags(16777216)
        got8 = step(Y, C, 0.1).astype(np.float64)
    finally:
        lib.wmf_debug_set_flags(0)
    rel_8, zero_8 = worst_row(got8, want)
    assert fro(got8, want) <= HALF_FRO and rel_8 <= HALF_ROW and zero_8 == 0.0, (fro(got8, want), rel_8)
    # the two kernels differ in arithmetic (split-f16 products with an LDS-DMA ring here, f32 MFMAs from a register ring
    # there): they agree to the accuracy either has against the oracle, far inside the row tolerance
    rel = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    record_error(f"heavy_rows_k128[bias={int(bias)}]", worst_row_vs_oracle=rel_o, fro_vs_oracle=fro(got, want),
                 split_f16_vs_f32_kernel_worst_row=rel.max())
    # (measured: 2.9e-6 worst row here, 3.5e-7 on cfg3's item side -- the one check that isolates the split-f16 arithmetic)
    assert rel.max() <= 8e-6, (rel.max(), int(rel.argmax()), int(deg[rel.argmax()]))
    Yt, bvec = Y.astype(np.float64).copy(), np.zeros(m_items)
    if bias:
        bvec, Yt[:, 0] = Yt[:, 0].copy(), 1.0
    G = Yt.T @ Yt + 0.1 * np.eye(Y.shape[1])
    for u in rng.choice(n, 40, replace=False):
        lo, hi = C.indptr[u], C.indptr[u + 1]
        U, w = Yt[C.indices[lo:hi]], C.data[lo:hi].astype(np.float64) - bvec[C.indices[lo:hi]]
        A, b = G + U.T @ (U * w[:, None]), (w + 1) @ U
        assert np.linalg.norm(A @ got[u] - b) <= 2e-5 * np.linalg.norm(b)


def _weight_range_matrix(n, m_items, rng, mode):
    """Every degree class (0, 1, 8, 9, 16, 17, 32, 33, 100, 400, 5000 = segments) under confidence weights far from the bench's
    10 log(1 + c) in [7, 19]: 'linear' pre-processing of large counts (alpha * c up to 1e6, wmf_model.py:122-123), tiny
    weights down to 1e-3, both in one row; mode 'overflow' adds entries whose sqrt(w) |v| is beyond the f16 range."""
    degs = [0, 1, 8, 9, 16, 17, 32, 33, 100, 400, 5000, 31, 64, 7] + [int(x) for x in rng.integers(1, 120, n - 14)]
    degs = [min(d, m_items) for d in degs]
    indptr = np.concatenate([[0], np.cumsum(degs)])
    indices = np.concatenate([np.sort(rng.choice(m_items, d, replace=False)) for d in degs if d]).astype(np.int32)
    z = int(indptr[-1])
    kind = rng.integers(0, 4, z)
    counts = np.where(kind == 0, rng.integers(1, 6, z), np.where(kind == 1, 10.0 ** rng.integers(1, 6, z), 1.0))
    w = np.where(kind == 0, 10 * np.log(1 + counts), 10.0 * counts)                      # 'log' next to 'linear'
    if mode != "bias":
        w = np.where(kind == 2, 10.0 ** rng.uniform(-3, 0, z), w)                        # tiny weights
    if mode == "overflow":
        hot = rng.random(z) < 0.01
        hot[indptr[8]] = hot[indptr[10] + 5] = True                                      # in a heavy and in a segmented row for sure
        w = np.where(hot, 10.0 ** rng.uniform(11, 13, z), w)
    return sp.csr_matrix((w.astype(np.float32), indices, indptr), shape=(n, m_items)), np.array(degs)


@pytest.mark.parametrize("k,bias,mode", [(64, False, "wide"), (128, True, "bias"), (256, False, "wide"), (256, True, "bias")])
def test_weight_range_with_float64_counts_meets_the_tolerance(WMF, k, bias, mode):
    """A float64 count matrix makes the reference solve its rows in float64 (wmf_model.py:237-239).  With weights up to 1e6 the
    float32 kernels are 1e-3 .. 6e-3 off per row (round 3's measurement), so such a matrix takes the float64 device path
    (recmodel_amd.wmf_model.F64_ROW_WEIGHT) and EVERY row meets the stated tolerance against the float64 oracle -- not
    "5 x NumPy-float32"."""
    rng = np.random.default_rng(1000 + k + bias)
    n, m_items = 300, 6000
    C, degs = _weight_range_matrix(n, m_items, rng, mode)
    C64 = as_f64(C)
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    Y = model.items.copy()
    if bias:
        Y[:, 0] *= 0.5
    want = (orc.recompute_factors_bias if bias else orc.recompute_factors)(Y, C64, 0.1, out_dtype="float64")
    got = (model.recompute_factors_bias if bias else model.recompute_factors)(Y, C64, 0.1)
    assert got.dtype == np.float32                                   # the reference stores the model's dtype (:217)
    den = np.linalg.norm(want, axis=1)
    ok = den > 0
    e_got = np.linalg.norm(got.astype(np.float64) - want, axis=1)[ok] / den[ok]
    record_error(f"weight_range_f64_counts[k={k},bias={int(bias)},{mode}]", worst_row=e_got.max(), median_row=float(np.median(e_got)))
    assert e_got.max() <= (HALF_ROW if k + bias <= 144 else WIDE_ROW), e_got.max()
    assert not got[degs == 0].any()


@pytest.mark.parametrize("k,bias,mode", [(64, False, "wide"), (64, True, "bias"), (128, False, "wide"), (128, True, "bias"),
                                         (256, False, "wide"), (256, True, "bias"),
                                         (64, False, "overflow"), (128, False, "overflow"), (128, True, "overflow"), (256, False, "overflow")])
def test_weight_range_of_the_class_surface(WMF, k, bias, mode):
    """The class surface admits any confidence weight (pre_process_count='linear' on raw counts, wmf_model.py:122-123), the
    split-f16 kernels scale operands by sqrt(w): weights from 1e-3 to 1e6 over every degree class must stay inside the stated
    tolerance wherever the reference's own float32 arithmetic does, and a row is never worse than 5 x what NumPy's float32
    restatement of the same row (the reference with a float32 count matrix) is against the float64 oracle.  'overflow': a few
    weights of 1e11 .. 1e13 whose scaled operands leave the f16 range -- those rows must come back finite and as good as
    that, through the pivoted kernel."""
    rng = np.random.default_rng(1000 + k + bias)
    n, m_items = 300, 6000
    C, degs = _weight_range_matrix(n, m_items, rng, mode)
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    Y = model.items.copy()
    if bias:
        Y[:, 0] *= 0.5                                                # weights >= 1 stay positive
    step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
    step_g = model.recompute_factors_bias if bias else model.recompute_factors
    want = step_o(Y, as_f64(C), 0.1, out_dtype="float64")
    ref32 = np.full(want.shape, np.inf)                                # the reference's arithmetic on float32 counts, row by row
    for r in range(n):                                                #   (a row whose float32 system is singular stays inf)
        one = sp.csr_matrix((C.data[C.indptr[r]: C.indptr[r + 1]], C.indices[C.indptr[r]: C.indptr[r + 1]],
                             [0, C.indptr[r + 1] - C.indptr[r]]), shape=(1, m_items))
        try:
            ref32[r] = step_o(Y, one, 0.1)[0]
        except np.linalg.LinAlgError:
            pass
    got = step_g(Y, C, 0.1).astype(np.float64)
    assert np.isfinite(got).all()
    den = np.linalg.norm(want, axis=1)
    ok = den > 0
    e_got = np.linalg.norm(got - want, axis=1)[ok] / den[ok]
    with np.errstate(invalid="ignore"):
        e_ref = np.nan_to_num(np.linalg.norm(ref32 - want, axis=1)[ok] / den[ok], nan=np.inf, posinf=np.inf)
    tol_row = HALF_ROW if k + bias <= 144 else WIDE_ROW
    bad = e_got > np.maximum(tol_row, 5 * e_ref)
    if mode == "overflow":                                            # weights of 1e11 .. 1e13: where NumPy's own float32 arithmetic is off by
        bad &= e_ref < 1e-2                                           # more than a percent nothing but finiteness (above) can be asked
    record_error(f"weight_range[k={k},bias={int(bias)},{mode}]", worst_row=e_got.max(), worst_row_numpy_f32=e_ref.max(),
                 median_row=float(np.median(e_got)), median_row_numpy_f32=float(np.median(e_ref)),
                 rows_above_tolerance=int((e_got > tol_row).sum()), rows_numpy_f32_above_tolerance=int((e_ref > tol_row).sum()))
    wmax = np.array([C.data[C.indptr[r]: C.indptr[r + 1]].max(initial=0.0) for r in range(n)])[ok]
    table = sorted(zip(degs[ok][bad].tolist(), wmax[bad].tolist(), e_got[bad].tolist(), e_ref[bad].tolist()))
    assert not bad.any(), (k, bias, mode, int(bad.sum()), [f"d={d} wmax={w_:.1e} gpu={a:.1e} numpy32={b:.1e}" for d, w_, a, b in table])
    assert not got[degs == 0].any()


@pytest.mark.parametrize("k", [16, 64, 128, 256])       # 128: the split-bf16 heavy-row kernel (a negative weight has no square root there)
def test_negative_weights_take_the_pivoted_path(WMF, k):
    """bias model with large fixed-side biases: w - bias < 0 for many entries, so A_u is not SPD
     and the rows must go through the LU kernel (LDS version for f <= 144, the
    global-workspace version above that)."""
    n, m_items = 400, 150
    C = ragged_matrix(n, m_items, seed=77)
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=True)
    Y = model.items.copy()
    Y[:, 0] = np.linspace(-5, 40, m_items)                            # weights are ~7..18: about half go negative
    want = orc.recompute_factors_bias(Y, as_f64(C), 0.1, out_dtype="float64")
    got = model.recompute_factors_bias(Y, C, 0.1)
    ok = np.linalg.norm(want, axis=1) < 1e3                           # skip rows that are themselves near singular
    assert ok.mean() > 0.9
    assert fro(got[ok], want[ok]) <= 1e-3
    assert not np.isnan(got).any()


def test_full_training_vs_oracle(WMF):
    from recmodel_amd import synth
    n, m_items, k = 1500, 400, 32
    ip, idx, val = synth.make_counts(n, m_items, 9, seed=4)
    counts = synth.to_scipy(ip, idx, val, (n, m_items)).astype(np.float64)
    for bias in (False, True):
        model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
        last = model.train(utility_mat=counts, iterations=3, eval_mat=counts, count_mat=counts, cores=2, stopping_rounds=5)
        assert model.users.dtype == np.float64 and model.items.dtype == np.float64    # the reference's cores > 1 quirk
        r_last, r_hist, r_users, r_items = orc.train(m_items, n, k, 0.1, counts, 3, counts, count_mat=counts,
                                                     weighted=True, bias=bias, stopping_rounds=5)
        assert last == r_last == 2
        assert fro(model.users, r_users) <= TRAIN_FRO and fro(model.items, r_items) <= TRAIN_FRO
        assert abs(model.eval_prec(counts) - r_hist[-1]) <= TRAIN_MSE * r_hist[-1]
        # top-10 overlap on sampled users
        cand = np.arange(m_items)
        overlap = []
        for u in range(0, n, 97):
            a = set(model.rank(cand, u, topn=10).tolist())
            b = set(orc.rank(r_users, r_items, cand, u, 10, bias).tolist())
            overlap.append(len(a & b) / 10)
        assert np.mean(overlap) >= 0.99


# ------------------------------------------------------------------ a5 / a6: the Pool variants in float64
def test_pool_variants_float64_vs_reference_golden_and_oracle(WMF):
    """recompute_factors_par / recompute_factors_bias_par (wmf_model.py:242-309) through wmf_recompute_factors_f64_host:
    float64 rows against the reference's own outputs (golden half_par.npz; its first Gramian is a float32 product, so 1e-5)
    and against the float64 oracle fed float64 inputs (the arithmetic itself: 1e-10)."""
    g = load_golden("half_par.npz")
    C = csr_from(g, "C")
    assert C.dtype == np.float64
    for bias in (False, True):
        Y, ref = g[f"items0_bias{int(bias)}"], g[f"users_par_bias{int(bias)}"]
        model = WMF(num_items=30, num_users=40, dim=8, gamma=0.1, weighted=True, bias=bias)
        fn = model.recompute_factors_bias_par if bias else model.recompute_factors_par
        Y_in = Y.copy()
        got = fn(Y_in, C, 0.1, cores=2)
        assert got.dtype == np.float64 and got.shape == ref.shape
        assert np.array_equal(Y_in, Y)                                   # the caller's factors are left alone
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6)
        step = orc.recompute_factors_bias if bias else orc.recompute_factors
        want = step(Y.astype(np.float64), C, 0.1, dtype="float64")
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13)
        empty = np.flatnonzero(np.diff(C.indptr) == 0)
        assert len(empty) and np.all(got[empty] == 0)                    # rows without stored entries: zeros (:274-276, :296-298)
        got32 = fn(Y, C.astype(np.float32), 0.1, cores=2)                # all-float32 inputs stay float32 in the reference
        assert got32.dtype == np.float32 and fro(got32, ref) <= HALF_FRO


@pytest.mark.parametrize("k,bias", [(16, False), (64, True), (128, True), (256, False)])
def test_float64_half_step_all_degree_classes(WMF, k, bias):
    """The float64 device path on the ragged matrix of the float32 tests (degrees 0 ... full row, a stored zero, a
    duplicate) at narrow and wide f, and -- bias -- with weights that go negative (LU with partial pivoting, as gesv)."""
    n, m_items = 120, 150
    C = as_f64(ragged_matrix(n, m_items, seed=5 + k))
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    Y = model.items.astype(np.float64)
    if bias:
        Y[:, 0] = np.linspace(-5, 30, m_items)                            # about a third of the weights go negative
    fn = model.recompute_factors_bias_par if bias else model.recompute_factors_par
    want = (orc.recompute_factors_bias if bias else orc.recompute_factors)(Y, C, 0.1, dtype="float64")
    ok = np.linalg.norm(want, axis=1) < 1e3                               # indefinite rows can be near singular themselves
    assert ok.mean() > 0.9
    from recmodel_amd import _lib
    lib = _lib.load()
    # rows with 1 .. 32 entries through the whitened low-rank form (most rows here, so it is on), then every row through the
    # direct f x f kernel (debug flag 134217728)
    for flags in (0, 134217728):
        try:
            lib.wmf_debug_set_flags(flags)
            got = fn(Y, C, 0.1)
        finally:
            lib.wmf_debug_set_flags(0)
        assert got.dtype == np.float64 and not np.isnan(got).any()
        record_error(f"float64_half_step[k={k},bias={int(bias)},lowrank={int(flags == 0)}]", fro=fro(got[ok], want[ok]))
        assert fro(got[ok], want[ok]) <= 1e-8
        assert np.all(got[0] == 0)                                        # the empty row


def test_train_cores2_float64_vs_reference_golden(WMF):
    """train(cores=2) on a float64 count matrix against the reference's own run (golden train_par.npz) and the oracle."""
    g = load_golden("train_par.npz")
    counts, util = csr_from(g, "counts"), csr_from(g, "util")
    for bias in (False, True):
        model = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=8, gamma=0.1, weighted=True, bias=bias)
        last = model.train(utility_mat=util, iterations=3, eval_mat=util, count_mat=counts, cores=2, stopping_rounds=5)
        assert last == int(g[f"last_iter_bias{int(bias)}"])
        assert model.users.dtype == np.float64 and model.items.dtype == np.float64
        assert fro(model.users, g[f"users_bias{int(bias)}"]) <= 2e-5 and fro(model.items, g[f"items_bias{int(bias)}"]) <= 2e-5
        assert abs(model.eval_prec(util) - float(g[f"mse_final_bias{int(bias)}"])) <= 1e-5 * float(g[f"mse_final_bias{int(bias)}"])
    # a float32 count matrix keeps the reference's Pool variants in float32
    model = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=8, gamma=0.1, weighted=True)
    model.train(utility_mat=util, iterations=1, eval_mat=util, count_mat=counts.astype(np.float32), cores=2, stopping_rounds=5)
    assert model.users.dtype == np.float32


def test_train_cores2_integer_counts_vs_reference_golden(WMF):
    """An int64 count matrix with cores = 2: float64 factors in the reference ('log': the transform is float64; 'linear': int64
    weights promote the row products) -- golden train_par_int.npz from the reference itself."""
    g = load_golden("train_par_int.npz")
    counts, util = csr_from(g, "counts"), csr_from(g, "util")
    assert counts.dtype == np.int64
    for mode in ("log", "linear"):
        model = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=6, gamma=0.1, weighted=True, bias=(mode == "linear"))
        last = model.train(utility_mat=util, iterations=2, eval_mat=util, count_mat=counts, cores=2, stopping_rounds=5,
                           pre_process_count=mode, alpha=(10 if mode == "log" else 2))
        assert last == int(g[f"last_iter_{mode}"])
        assert model.users.dtype == np.float64 and model.items.dtype == np.float64
        # (2e-5, not 1e-8: the reference's first Gramian is a float32 product of the float32 initial items, ours is float64.  That
        # gate alone would let a float32 solve cast to float64 through, so: the values must not be float32 numbers)
        assert fro(model.users, g[f"users_{mode}"]) <= 2e-5 and fro(model.items, g[f"items_{mode}"]) <= 2e-5
        assert (model.users != model.users.astype(np.float32)).mean() > 0.9 and (model.items != model.items.astype(np.float32)).mean() > 0.9
        assert abs(model.eval_prec(util) - float(g[f"mse_final_{mode}"])) <= 1e-5 * float(g[f"mse_final_{mode}"])
    # cores = 1 on the same matrix: float32 factors (recompute_factors casts each row back, :217, :239)
    model = WMF(num_items=util.shape[1], num_users=util.shape[0], dim=6, gamma=0.1, weighted=True)
    model.train(utility_mat=util, iterations=1, eval_mat=util, count_mat=counts, cores=1, stopping_rounds=5)
    assert model.users.dtype == np.float32


def test_train_dtype_follows_numpy_promotion_vs_reference_golden(WMF):
    """Which dtype the factors end in is NumPy's promotion of the model dtype and the TRANSFORMED counts (golden
    train_dtypes.npz from the reference itself, round 4): int16 counts with cores = 2 stay float32 (np.log of int16 is float32,
    float32 * int16 is float32), a float64 model on float32 counts is float64 whatever `cores` is."""
    g = load_golden("train_dtypes.npz")
    base, util = csr_from(g, "counts"), csr_from(g, "util")
    cases = {"int16_log": (np.int16, "log", "float32", 2, 10), "int16_linear": (np.int16, "linear", "float32", 2, 2),
             "f32_model64": (np.float32, "log", "float64", 2, 10), "f32_model64_c1": (np.float32, "log", "float64", 1, 10)}
    for name, (cdt, mode, mdt, cores, alpha) in cases.items():
        counts = sp.csr_matrix((np.rint(base.data).astype(cdt), base.indices, base.indptr), shape=base.shape)
        model = WMF(num_items=40, num_users=80, dim=5, gamma=0.1, weighted=True, bias=False, dtype=mdt)
        last = model.train(utility_mat=util, iterations=2, eval_mat=util, count_mat=counts, cores=cores, stopping_rounds=5,
                           pre_process_count=mode, alpha=alpha)
        want_u, want_i = g[f"users_{name}"], g[f"items_{name}"]
        assert last == int(g[f"last_iter_{name}"]), name
        assert model.users.dtype == want_u.dtype and model.items.dtype == want_i.dtype, (name, model.users.dtype, want_u.dtype)
        tol = 1e-8 if want_u.dtype == np.float64 else 2e-4
        assert fro(model.users, want_u) <= tol and fro(model.items, want_i) <= tol, (name, fro(model.users, want_u), fro(model.items, want_i))
        assert abs(model.eval_prec(util) - float(g[f"mse_final_{name}"])) <= 1e-4 * float(g[f"mse_final_{name}"]), name


def test_train_argument_errors_like_reference(WMF):
    c = sp.random(30, 20, density=0.3, format="csr", random_state=1)
    m = WMF(num_items=20, num_users=30, dim=4, gamma=0.1, weighted=True)
    with pytest.raises(ValueError):
        m.train(utility_mat=c, iterations=1, eval_mat=c, count_mat=c, pre_process_count="sqrt")
    with pytest.raises(ValueError):
        m.train(utility_mat=c, iterations=1, eval_mat=c, count_mat=c, cores=0)
    with pytest.raises(AttributeError):
        m.train(utility_mat=c, iterations=1, eval_mat=None, count_mat=c)      # wmf_model.py:61-63


# ------------------------------------------------------------------ size-independent properties at bench scale
def test_properties_at_benchmark_scale():
    """cfg2-shaped matrix (k=64, ~2 M stored entries here; the bench runs the 20 M version): every
    updated row must satisfy its own normal equations, empty rows must be zero, and the weighted
    loss must not increase across half steps."""
    from recmodel_amd import synth
    from recmodel_amd.engine import AlsEngine
    n_users, n_items, k = 100_000, 10_000, 64
    ip, idx, val = synth.make_counts(n_users, n_items, 20, seed=1995, device="cuda")
    val = 10 * torch.log(1 + val)
    eng = AlsEngine(n_users, n_items, k, False, 0.1)
    eng.set_interactions(ip, idx, val)
    eng.set_factors("items", orc.init_items(n_items, k))
    C = synth.to_scipy(ip, idx, val, (n_users, n_items))

    def weighted_loss(X, Y):
        # sum_ui c_ui (p_ui - x.y)^2 over stored entries + implicit zeros + ridge (Hu/Koren/Volinsky objective)
        rows = np.repeat(np.arange(n_users), np.diff(C.indptr))
        s = np.einsum("ij,ij->i", X[rows], Y[C.indices])
        stored = ((C.data + 1) * (1 - s) ** 2 - s ** 2).sum()
        all_pairs = np.trace((X.T @ X) @ (Y.T @ Y))
        return stored + all_pairs + 0.1 * ((X ** 2).sum() + (Y ** 2).sum())

    losses = []
    for it in range(2):
        eng.half_step("users")
        X, Y = eng.get_factors("users").astype(np.float64), eng.get_factors("items").astype(np.float64)
        losses.append(weighted_loss(X, Y))
        eng.half_step("items")
        X, Y = eng.get_factors("users").astype(np.float64), eng.get_factors("items").astype(np.float64)
        losses.append(weighted_loss(X, Y))
    eng.check_numerics()
    assert all(b <= a * (1 + 1e-6) for a, b in zip(losses, losses[1:])), losses
    # normal-equation residual of sampled item rows against the users that produced them
    CT = C.T.tocsr()
    G = X.T @ X + 0.1 * np.eye(k)
    worst = 0.0
    for i in np.random.default_rng(0).choice(n_items, 200, replace=False):
        lo, hi = CT.indptr[i], CT.indptr[i + 1]
        U, w = X[CT.indices[lo:hi]], CT.data[lo:hi].astype(np.float64)
        A, b = G + U.T @ (U * w[:, None]), (w + 1) @ U
        worst = max(worst, np.linalg.norm(A @ Y[i] - b) / max(np.linalg.norm(b), 1e-30))
    assert worst <= 2e-4, worst


@pytest.mark.parametrize("k,bias", [(64, False), (33, True)])
def test_chunked_solve_is_bit_identical(k, bias):
    """Multi-rank runs solve their rows in chunks (one row plan per chunk) so that the exchange of a chunk
    overlaps the solve of the next; rows are independent, so chunking must not change a single bit --
    including rows that are split into segments."""
    from recmodel_amd import synth
    from recmodel_amd.engine import AlsEngine
    n_users, n_items = 30_011, 1_003
    ip, idx, val = synth.make_counts(n_users, n_items, 25, seed=77, device="cuda", zipf_a=1.05)
    val = 10 * torch.log(1 + val)
    out = []
    for chunks in (1, 5):
        eng = AlsEngine(n_users, n_items, k, bias, 0.1, chunks=chunks)
        eng.set_interactions(ip, idx, val)
        assert len(eng.csr_chunks["items"]) == chunks
        eng.set_factors("items", orc.init_items(n_items, k, bias))
        eng.half_step("users")
        eng.half_step("items")
        eng.check_numerics()
        out.append((eng.get_factors("users"), eng.get_factors("items")))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("k,bias", [(16, False), (64, False), (64, True), (50, False), (128, True), (128, False)])
def test_accumulate_then_eliminate_equals_the_row_solve(k, bias):
    """The reduce-scatter exchange splits the heavy-row kernel at the sum over stored entries: partial systems of
    row subsets (here: two halves of every row's entries, as two ranks would hold them), added, then eliminated, must
    give what wmf_solve_rows gives on the whole rows -- for every degree class, including rows without entries."""
    from recmodel_amd.engine import AlsEngine, _ptr, _stream
    C = ragged_matrix(700, 300, seed=k + 5 * bias)
    eng = AlsEngine(700, 300, k, bias, 0.1)
    ip = torch.from_numpy(C.indptr.astype(np.int64)).cuda()
    eng.set_interactions(ip, torch.from_numpy(C.indices.astype(np.int64)).cuda(), torch.from_numpy(C.data).cuda())
    eng.set_factors("items", orc.init_items(300, k, bias))
    eng.half_step("users")
    eng.check_numerics()
    want = eng.g["users"].clone()                                  # whitened solutions of the ordinary path
    K, f, ld = eng.K, eng.f, eng.ld
    pr = K.partial_row_floats(f)
    assert pr > 0
    full = eng.csr["users"]
    total = torch.zeros(700, pr, device="cuda")
    cols = full.indices.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(700, device="cuda"), full.indptr[1:] - full.indptr[:-1])
    for half in (0, 1):                                            # entries with even / odd column: two "ranks"
        m = (cols % 2) == half
        cnt = torch.bincount(rows[m], minlength=700)
        ptr = torch.zeros(701, dtype=torch.int64, device="cuda")
        torch.cumsum(cnt, 0, out=ptr[1:])
        idx, val = full.indices[m].contiguous(), full.values[m].contiguous()
        part = torch.full((700, pr), 7.0, device="cuda")
        w_eff = torch.empty_like(val) if bias and not eng.split else None    # (split layout: no bias pass, no workspace)
        K.accumulate_rows(eng.V["items"], eng.bias_vec["items"] if bias else None, ptr, cnt.to(torch.int32), idx, val, 700,
                          idx.numel(), f, ld, part, w_eff)
        total += part
    g2 = torch.zeros(700, ld, device="cuda")
    fail = torch.zeros(4, dtype=torch.int32, device="cuda")
    K.eliminate_rows(total, 700, f, ld, g2, fail, torch.zeros(700, dtype=torch.int32, device="cuda"))
    assert int(fail[0]) == 0
    a, b = g2.cpu().numpy().astype(np.float64), want.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(a - b) <= 2e-5 * np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)
    assert not a[0].any()                                          # the row without entries


@pytest.mark.parametrize("m,f,bias", [(200_003, 129, 1), (300_001, 128, 0), (140_000, 64, 0), (131, 129, 1)])
def test_gramian_at_wave_count_boundaries(m, f, bias):
    """The Gramian kernels deal 32-row chunks (wide factors) or row ranges over a wave count that is rounded to a whole number of
    waves per SIMD above 1024: sizes just past that rounding, a ragged last chunk, and fewer rows than one chunk per wave."""
    from recmodel_amd import _lib
    from recmodel_amd.engine import _ptr, _stream
    lib = _lib.load()
    ld = lib.wmf_ld_for(f)
    g = torch.Generator(device="cuda").manual_seed(m)
    Yd = torch.zeros(m, ld, device="cuda")
    Yd[:, :f] = torch.randn(m, f, device="cuda", generator=g)
    ws = torch.empty(int(lib.wmf_gram_workspace_bytes(f)), dtype=torch.uint8, device="cuda")
    G = torch.zeros(f * f, dtype=torch.float64, device="cuda")
    _lib.check(lib.wmf_gram(_ptr(Yd), m, f, ld, bias, _ptr(G), _ptr(ws), _stream()))
    Yt = Yd[:, :f].double()
    if bias:
        Yt[:, 0] = 1
    Gref = (Yt.T @ Yt).cpu().numpy()
    np.testing.assert_allclose(G.cpu().numpy().reshape(f, f), Gref, rtol=0, atol=2e-6 * np.abs(Gref).max())


def test_device_building_blocks_individually():
    """gram / factorize / row_transform against NumPy, including a non-positive-definite Gramian."""
    from recmodel_amd import _lib
    from recmodel_amd.engine import _ptr, _stream
    lib = _lib.load()
    rng = np.random.default_rng(5)
    for f, bias in ((7, 0), (64, 0), (129, 1), (200, 0), (257, 1)):
        ld = lib.wmf_ld_for(f)
        m = 1234
        Y = rng.standard_normal((m, f)).astype(np.float32)
        Yd = torch.zeros(m, ld, device="cuda"); Yd[:, :f] = torch.from_numpy(Y).cuda()
        ws = torch.empty(int(lib.wmf_gram_workspace_bytes(f)), dtype=torch.uint8, device="cuda")
        G = torch.zeros(f * f, dtype=torch.float64, device="cuda")
        _lib.check(lib.wmf_gram(_ptr(Yd), m, f, ld, bias, _ptr(G), _ptr(ws), _stream()))
        Yt = Y.astype(np.float64).copy()
        if bias:
            Yt[:, 0] = 1
        Gref = Yt.T @ Yt
        np.testing.assert_allclose(G.cpu().numpy().reshape(f, f), Gref, rtol=0, atol=2e-6 * np.abs(Gref).max())
        Ww, Wu = torch.zeros(f, ld, device="cuda"), torch.zeros(f, ld, device="cuda")
        info = torch.zeros(4, dtype=torch.int32, device="cuda")
        _lib.check(lib.wmf_factorize(_ptr(G), f, ld, 0.1, _ptr(Ww), _ptr(Wu), _ptr(info), _ptr(ws), _stream()))
        assert int(info[0]) == 0
        Linv = np.linalg.inv(np.linalg.cholesky(G.cpu().numpy().reshape(f, f) + 0.1 * np.eye(f)))
        np.testing.assert_allclose(Wu.cpu().numpy()[:, :f], Linv, rtol=0, atol=1e-6 * np.abs(Linv).max())
        np.testing.assert_allclose(Ww.cpu().numpy()[:, :f], Linv.T, rtol=0, atol=1e-6 * np.abs(Linv).max())
        Vref = Yt @ Linv.T
        tol = 3e-6 * np.abs(Vref).max() * np.sqrt(f)
        ldv = lib.wmf_whitened_row_floats(f, ld, bias)
        assert ldv == (128 if (f, bias) == (129, 1) else ld)
        if ldv != ld:
            # f = 16 m + 1 <= 144 with biases: the SPLIT LAYOUT (include/wmf_hip.h, wmf_row_transform) -- a packed body of
            # f - 1 floats per row and the pairs {feature f - 1, bias}; nothing is written behind either array
            V = torch.full((m * ldv + 8,), 3.0, device="cuda")
            bv = torch.full((2 * m + 8,), 5.0, device="cuda")
            _lib.check(lib.wmf_row_transform(_ptr(Yd), m, f, ld, _ptr(Ww), bias, _ptr(V), _ptr(bv), _stream()))
            Vn, bn = V.cpu().numpy(), bv.cpu().numpy()
            np.testing.assert_allclose(Vn[: m * ldv].reshape(m, ldv), Vref[:, : f - 1], rtol=0, atol=tol)
            np.testing.assert_allclose(bn[: 2 * m].reshape(m, 2)[:, 0], Vref[:, f - 1], rtol=0, atol=tol)
            np.testing.assert_array_equal(bn[: 2 * m].reshape(m, 2)[:, 1], Y[:, 0])
            assert np.all(Vn[m * ldv:] == 3.0) and np.all(bn[2 * m:] == 5.0)
            with pytest.raises((ValueError, _lib.WmfLibraryError)):   # the pairs are not optional there
                _lib.check(lib.wmf_row_transform(_ptr(Yd), m, f, ld, _ptr(Ww), bias, _ptr(V), None, _stream()))
            continue
        V = torch.full((m, ld), 3.0, device="cuda")
        bv = torch.zeros(m, device="cuda")
        _lib.check(lib.wmf_row_transform(_ptr(Yd), m, f, ld, _ptr(Ww), bias, _ptr(V), _ptr(bv) if bias else None, _stream()))
        np.testing.assert_allclose(V.cpu().numpy()[:, :f], Vref, rtol=0, atol=tol)
        assert np.all(V.cpu().numpy()[:, f:] == 0)         # (f = 257 is beyond the split widths: a bias vector only)
        if bias:
            np.testing.assert_array_equal(bv.cpu().numpy(), Y[:, 0])
    # not positive definite -> info > 0, no exception inside the kernel, zero transforms
    f, ld = 8, 8
    G = torch.from_numpy(-np.eye(f).reshape(-1)).cuda()
    Ww, Wu = torch.ones(f, ld, device="cuda"), torch.ones(f, ld, device="cuda")
    info = torch.zeros(4, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(lib.wmf_gram_workspace_bytes(f)), dtype=torch.uint8, device="cuda")
    _lib.check(lib.wmf_factorize(_ptr(G), f, ld, 0.1, _ptr(Ww), _ptr(Wu), _ptr(info), _ptr(ws), _stream()))
    assert int(info[0]) == 1 and float(Ww.abs().sum()) == 0.0


@pytest.mark.parametrize("k,bias", [(32, False), (128, True), (256, False), (256, True), (150, False), (176, True)])
def test_very_heavy_rows_are_split_into_segments(WMF, k, bias):
    """Rows with more than 4096 stored entries (power-law heads) are accumulated by
    several waves (f <= 144) or workgroups (the four-waves-per-row kernel beyond) in 2048-entry segments and
    combined in a fixed order."""
    rng = np.random.default_rng(9)
    n, m_items = 60, 13000
    degs = [5000, 9000, 13000, 4096, 4097] + [int(x) for x in rng.integers(1, 200, n - 5)]
    indptr = np.concatenate([[0], np.cumsum(degs)])
    indices = np.concatenate([np.sort(rng.choice(m_items, d, replace=False)) for d in degs]).astype(np.int32)
    data = (10 * np.log(1 + rng.integers(1, 6, indptr[-1]))).astype(np.float32)
    C = sp.csr_matrix((data, indices, indptr), shape=(n, m_items))
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
    step_g = model.recompute_factors_bias if bias else model.recompute_factors
    want = step_o(model.items, as_f64(C), 0.1, out_dtype="float64")
    got = step_g(model.items, C, 0.1)
    rel, _ = worst_row(got, want)
    tol_fro, tol_row = (HALF_FRO, HALF_ROW) if k + bias <= 144 else (WIDE_FRO, WIDE_ROW)
    assert fro(got, want) <= tol_fro and rel <= tol_row, (fro(got, want), rel)
    assert np.array_equal(got, step_g(model.items, C, 0.1))          # fixed combination order: bitwise reproducible


@pytest.mark.parametrize("f,bias", [(193, True), (209, True), (225, True), (241, True), (257, True), (225, False), (240, False)])
def test_wide_rows_under_full_occupancy_are_right_and_reproducible(WMF, f, bias):
    """144 < f <= 257 with EVERY row heavy (30 .. 700 entries), more rows than the chip holds workgroups: all resident
    workgroups of the four-waves-per-row kernel are busy at once.  Round 3's fuzzing found the border widths with 13 .. 15 blocks
    (f = 209, 225, 241) wrong and different from run to run in exactly this situation -- two such workgroups on one CU -- while
    the small ragged matrices of the other tests never filled a CU.  Round 4: the staging of wmf_rowsplit.hip is branch-free (the
    divergent region `lane < 4 NFB` those three widths alone had is gone) and they run two workgroups per CU again; the same test
    passes with every round-3 guard compiled out (tools/lab/r4_rs2.sh).  Two runs must agree bit for bit, and 120 sampled rows
    must match the float64 oracle."""
    rng = np.random.default_rng(f)
    n, m_items = 1200, 700
    k = f - int(bias)
    deg = rng.integers(30, 700, n)
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([np.sort(rng.choice(m_items, d, replace=False)) for d in deg]).astype(np.int32)
    data = (10 * np.log(1 + rng.integers(1, 8, indptr[-1]))).astype(np.float32)
    C = sp.csr_matrix((data, indices, indptr), shape=(n, m_items))
    model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    Y = model.items.copy()
    if bias:
        Y[:, 0] *= 0.5
    step_g = model.recompute_factors_bias if bias else model.recompute_factors
    got = step_g(Y, C, 0.1)
    again = step_g(Y, C, 0.1)
    assert np.array_equal(got, again), int((np.abs(got - again).max(axis=1) > 0).sum())
    rows = rng.choice(n, 120, replace=False)
    sub = sp.csr_matrix((np.concatenate([data[indptr[r]: indptr[r + 1]] for r in rows]).astype(np.float64),
                         np.concatenate([indices[indptr[r]: indptr[r + 1]] for r in rows]),
                         np.concatenate([[0], np.cumsum(deg[rows])])), shape=(len(rows), m_items))
    want = (orc.recompute_factors_bias if bias else orc.recompute_factors)(Y, sub, 0.1, out_dtype="float64")
    rel, _ = worst_row(got[rows], want)
    record_error(f"wide_rows_full_occupancy[f={f},bias={int(bias)}]", worst_row=rel, fro=fro(got[rows], want))
    assert fro(got[rows], want) <= WIDE_FRO and rel <= WIDE_ROW, (fro(got[rows], want), rel)


def test_wide_rows_reproducible_at_every_block_count(WMF):
    """Two runs of the four-waves-per-row kernel must agree bit for bit at every block count 10 .. 16, border or not, with all
    resident workgroups busy (the check that would have caught round 3's f = 209 / 225 / 241 finding in round 2)."""
    rng = np.random.default_rng(77)
    n, m_items = 1100, 600
    deg = rng.integers(33, 600, n)
    indptr = np.concatenate([[0], np.cumsum(deg)])
    indices = np.concatenate([np.sort(rng.choice(m_items, d, replace=False)) for d in deg]).astype(np.int32)
    data = (10 * np.log(1 + rng.integers(1, 8, indptr[-1]))).astype(np.float32)
    C = sp.csr_matrix((data, indices, indptr), shape=(n, m_items))
    for f in (145, 160, 161, 176, 177, 192, 193, 208, 209, 224, 225, 240, 241, 256, 257):
        model = WMF(num_items=m_items, num_users=n, dim=f, gamma=0.1, weighted=True, bias=False)
        got = model.recompute_factors(model.items, C, 0.1)
        again = model.recompute_factors(model.items, C, 0.1)
        assert np.isfinite(got).all(), f
        assert np.array_equal(got, again), (f, int((np.abs(got - again).max(axis=1) > 0).sum()))


# ------------------------------------------------------------------ degenerate shapes
@pytest.mark.parametrize("bias", [False, True])
def test_degenerate_shapes(WMF, bias):
    """Nothing stored at all, a single user, a single item, k = 1: the reference handles them (solve(G, 0) = 0,
    1 x 1 systems), so must the kernels."""
    step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
    rng = np.random.default_rng(3)
    for n, m_items, k, dens in ((5, 7, 4, 0.0), (1, 9, 3, 0.6), (6, 1, 2, 0.7), (40, 30, 1, 0.3), (3, 200, 16, 1.0)):
        C = sp.random(n, m_items, density=dens, random_state=int(rng.integers(1 << 30)), format="csr", dtype=np.float32)
        C.data = (1 + 10 * C.data).astype(np.float32)
        model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
        step_g = model.recompute_factors_bias if bias else model.recompute_factors
        want = step_o(model.items, as_f64(C), 0.1, out_dtype="float64")
        got = step_g(model.items, C, 0.1)
        assert got.shape == want.shape and got.dtype == np.float32
        assert fro(got, want) <= HALF_FRO or np.abs(got - want).max() <= 1e-6, (n, m_items, k, dens, fro(got, want))
        if C.nnz == 0:
            assert not got.any()
    # a full training run on a matrix with an empty user, an empty item and one dense row
    C = sp.random(30, 20, density=0.2, random_state=7, format="lil", dtype=np.float64)
    C[9, :] = 3
    C[4, :] = 0
    C[:, 6] = 0
    C = sp.csr_matrix(C)
    C.data = np.ceil(C.data * 4)
    model = WMF(num_items=20, num_users=30, dim=5, gamma=0.1, weighted=True, bias=bias)
    last = model.train(utility_mat=C.copy(), count_mat=C.copy(), iterations=3, eval_mat=C.copy(), stopping_rounds=5)
    o_last, _, o_users, o_items = orc.train(20, 30, 5, 0.1, C.copy(), 3, C.copy(), count_mat=C.copy(), bias=bias, stopping_rounds=5)
    assert last == o_last
    assert fro(model.users, o_users.astype(np.float64)) <= TRAIN_FRO and fro(model.items, o_items.astype(np.float64)) <= TRAIN_FRO
    if not bias:
        assert not model.users[4].any() and not model.items[6].any()     # rows without stored entries are exactly zero
