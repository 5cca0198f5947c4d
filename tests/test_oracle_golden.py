"""The oracle (oracle/wmf_oracle.py) against golden vectors captured from the reference WMF class
(tests/golden/make_golden.py).  CPU only.  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import csr_from, load_golden
from oracle import wmf_oracle as orc

# The oracle repeats the reference's NumPy calls, so on the same BLAS it agrees to the last bit;
# the tolerance only leaves room for a different BLAS/LAPACK build on the GPU box.
RTOL, ATOL = 2e-5, 2e-6


def test_init_items_matches_reference_rng():
    g = load_golden("init.npz")
    np.testing.assert_array_equal(orc.init_items(37, 6, False, 1993), g["items_bias0"])
    np.testing.assert_array_equal(orc.init_items(37, 6, True, 1993), g["items_bias1"])
    np.testing.assert_array_equal(orc.init_items(9, 4, False, 7), g["items_seed7"])
    assert g["items_bias1"].shape == (37, 7) and g["items_bias0"].dtype == np.float32


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("cdt", ["float32", "float64"])
def test_half_steps(bias, cdt):
    g = load_golden(f"half_bias{int(bias)}_{cdt}.npz")
    C, CT = csr_from(g, "C"), csr_from(g, "CT")
    assert C.data.dtype == np.dtype(cdt)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    users1 = step(g["items0"], C, float(g["gamma"]))
    np.testing.assert_allclose(users1, g["users1"], rtol=RTOL, atol=ATOL)
    assert users1.dtype == np.float32
    items1 = step(g["users1"], CT, float(g["gamma"]))
    np.testing.assert_allclose(items1, g["items1"], rtol=RTOL, atol=ATOL)
    users2 = step(g["items1"], C, float(g["gamma"]))
    np.testing.assert_allclose(users2, g["users2"], rtol=RTOL, atol=ATOL)
    # edge cases baked into the fixture: empty user row -> zeros; never-chosen item -> zeros
    assert np.all(g["users1"][3] == 0) and np.all(users1[3] == 0)
    assert np.all(g["items1"][11] == 0) and np.all(items1[11] == 0)
    # confidence transform of the raw counts gives C (train() defaults)
    counts = csr_from(g, "counts")
    np.testing.assert_allclose(orc.confidence_transform(counts.data), C.data, rtol=1e-6)


def test_pool_variants_are_float64_and_equal_serial():
    g = load_golden("half_par.npz")
    C = csr_from(g, "C")
    for bias in (False, True):
        ref = g[f"users_par_bias{int(bias)}"]
        assert ref.dtype == np.float64
        step = orc.recompute_factors_bias if bias else orc.recompute_factors
        got = step(g[f"items0_bias{int(bias)}"], C, 0.1, out_dtype="float64")
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-12)


def test_train_with_pool_variants_keeps_float64_factors():
    """train(cores=2) on a float64 count matrix: golden from the reference itself (tests/golden/make_golden.py train_par)."""
    g = load_golden("train_par.npz")
    counts, util = csr_from(g, "counts"), csr_from(g, "util")
    for bias in (False, True):
        last, hist, users, items = orc.train(
            num_items=util.shape[1], num_users=util.shape[0], dim=8, gamma=0.1, utility_mat=util, iterations=3,
            eval_mat=util, count_mat=counts, weighted=True, bias=bias, stopping_rounds=5, cores=2)
        assert users.dtype == np.float64 and items.dtype == np.float64
        assert last == int(g[f"last_iter_bias{int(bias)}"])
        # the first Gramian is taken of float32 factors in float32 (BLAS order unknown): 1e-6, not 1e-12
        np.testing.assert_allclose(users, g[f"users_bias{int(bias)}"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(items, g[f"items_bias{int(bias)}"], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(hist[-1], g[f"mse_final_bias{int(bias)}"], rtol=1e-6)


@pytest.mark.parametrize("tag,bias", [("run", False), ("stop", False), ("bias", True)])
def test_train_control_flow_predict_rank(tag, bias):
    g = load_golden(f"train_{tag}.npz")
    counts, util = csr_from(g, "counts"), csr_from(g, "util")
    last, hist, users, items = orc.train(
        num_items=util.shape[1], num_users=util.shape[0], dim=int(g["dim"]), gamma=0.1,
        utility_mat=util, iterations=int(g["iterations"]), eval_mat=util, count_mat=counts,
        weighted=True, bias=bias, stopping_rounds=int(g["stopping_rounds"]),
        min_improvement=float(g["min_improvement"]))
    assert last == int(g["last_iter"])
    if tag == "stop":
        assert last < int(g["iterations"]) - 1          # the fixture does stop early
    np.testing.assert_allclose(hist, g["mse"], rtol=1e-5)
    np.testing.assert_allclose(users, g["users"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(items, g["items"], rtol=1e-4, atol=1e-5)
    # a7 / a8 / a10 on the reference's own final factors
    U, I = g["users"], g["items"]
    np.testing.assert_allclose(orc.predict(U, I, g["pred_users"], g["pred_items"], bias), g["pred"], rtol=1e-6)
    np.testing.assert_allclose(orc.predict(U, I, np.array([7]), g["rank_cand"], bias), g["pred_one_user"], rtol=1e-6)
    np.testing.assert_allclose(orc.eval_prec(U, I, util, bias), g["mse_final"], rtol=1e-6)
    np.testing.assert_allclose(orc.eval_prec(U, I, util, bias, "rmse"), g["rmse_final"], rtol=1e-6)
    np.testing.assert_allclose(orc.eval_prec(U, I, util, bias, "mae"), g["mae_final"], rtol=1e-6)
    cand = g["rank_cand"]
    np.testing.assert_array_equal(orc.rank(U, I, cand, 7, 5, bias), g["rank_top5"])      # argpartition side
    np.testing.assert_array_equal(orc.rank(U, I, cand, 7, 40, bias), g["rank_top40"])    # argsort side
    np.testing.assert_array_equal(orc.rank(U, I, cand, 7, None, bias), g["rank_all"])
    np.testing.assert_array_equal(np.array([orc.rank(U, I, cand, u, 3, bias) for u in (1, 2)]), g["rank_list0"])


def test_predict_length_mismatch_raises():
    U = np.ones((4, 3), np.float32)
    with pytest.raises(ValueError):
        orc.predict(U, U, np.array([0, 1]), np.array([0, 1, 2]))


def test_unweighted_branch():
    g = load_golden("train_unweighted.npz")
    util = csr_from(g, "util")
    last, hist, users, items = orc.train(
        num_items=util.shape[1], num_users=util.shape[0], dim=10, gamma=0.5, utility_mat=util,
        iterations=2, eval_mat=util, weighted=None, stopping_rounds=5)
    assert last == int(g["last_iter"]) and users.dtype == np.float64
    np.testing.assert_allclose(users, g["users"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(items, g["items"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(hist[-1], g["mse_final"], rtol=1e-9)


def test_bad_confidence_mode_raises():
    with pytest.raises(ValueError):
        orc.confidence_transform(np.ones(3), mode="sqrt")


@pytest.mark.parametrize("bias", [False, True])
def test_c_restatement_matches_reference_golden(bias):
    """oracle/wmf_oracle.c (double precision, own LU) against the reference's float64-count vectors."""
    import subprocess
    from conftest import ROOT
    subprocess.check_call(["make", "-s", "-C", f"{ROOT}/oracle"])
    from oracle import c_oracle
    g = load_golden(f"half_bias{int(bias)}_float64.npz")
    C, CT = csr_from(g, "C"), csr_from(g, "CT")
    # the reference forms its Gramian in float32 (wmf_model.py:215); with biases the initial Gramian is
    # ill conditioned (cond ~ 1e4) and that float32 rounding alone moves the reference by ~1e-5
    rtol, atol = (1e-4, 3e-5) if bias else (2e-5, 2e-6)
    np.testing.assert_allclose(c_oracle.half_step(g["items0"], C, 0.1, bias), g["users1"], rtol=rtol, atol=atol)
    np.testing.assert_allclose(c_oracle.half_step(g["users1"], CT, 0.1, bias), g["items1"], rtol=rtol, atol=atol)
    # and against the NumPy oracle (which, like the reference, forms the Gramian in float32: wmf_model.py:215)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    np.testing.assert_allclose(c_oracle.half_step(g["items0"], C, 0.1, bias), step(g["items0"], C, 0.1, out_dtype="float64"),
                               rtol=rtol, atol=max(atol, 5e-6))


@pytest.mark.parametrize("bias", [False, True])
def test_eval_topn_matches_reference_golden(bias):
    """Seeded Recall@N (base_model.py:100-148): same np.random draws, same ranking, same hit counts."""
    g = load_golden(f"eval_topn_bias{int(bias)}.npz")
    test = csr_from(g, "test")
    res, hits = orc.eval_topn(g["users"], g["items"], test, g["topn"], rand_sampled=int(g["rand_sampled"]),
                              random_state=int(g["random_state"]), bias=bias, return_hits=True)
    got = np.array([res[f"Recall@{n}"] for n in g["topn"]], dtype=np.float64)
    np.testing.assert_array_equal(got, g["recall"])
    assert hits.sum() > 0 and (np.diff(hits) >= 0).all()          # Recall@1 <= Recall@5 <= Recall@10
    with pytest.raises(ValueError):
        orc.eval_topn(g["users"], g["items"], test, [1, 5], rand_sampled=40)


def test_solve_row_in_slabs_equals_solve_row():
    """The slab form used for power-law heads at full size (tests/test_gpu_scale.py) is solve_row's arithmetic."""
    rng = np.random.default_rng(3)
    Y = rng.random((500, 12))
    G = orc.gramian(Y, 0.1, "float64")
    idx = rng.choice(500, 333, replace=False)
    w = rng.random(333) * 20
    want = orc.solve_row(G, Y, idx, w)
    slabs = [(Y[idx[lo: lo + 100]], w[lo: lo + 100]) for lo in range(0, 333, 100)]
    np.testing.assert_allclose(orc.solve_row_in_slabs(G, slabs), want, rtol=1e-12, atol=1e-14)


def test_train_cores2_on_integer_counts_is_float64():
    """Golden from the reference (make_golden.py train_par_int): an int64 count matrix with cores = 2 leaves float64 factors,
    with 'log' (the transform itself is float64) and with 'linear' (int64 weights promote the row products)."""
    g = load_golden("train_par_int.npz")
    counts, util = csr_from(g, "counts"), csr_from(g, "util")
    assert counts.dtype == np.int64
    for mode in ("log", "linear"):
        last, hist, users, items = orc.train(50, 90, 6, 0.1, util, 2, util, count_mat=counts, weighted=True, bias=(mode == "linear"),
                                             stopping_rounds=5, cores=2, pre_process_count=mode, alpha=(10 if mode == "log" else 2))
        assert last == int(g[f"last_iter_{mode}"]) and users.dtype == np.float64 and items.dtype == np.float64
        np.testing.assert_allclose(users, g[f"users_{mode}"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(items, g[f"items_{mode}"], rtol=RTOL, atol=ATOL)
        assert abs(hist[-1] - float(g[f"mse_final_{mode}"])) <= 1e-6 * float(g[f"mse_final_{mode}"])


def test_train_dtype_follows_numpy_promotion():
    """Golden from the reference (make_golden.py train_dtypes, round 4): int16 counts with cores = 2 stay float32, a float64
    model on float32 counts ends in float64 with cores = 1 and cores = 2."""
    import scipy.sparse as sp
    g = load_golden("train_dtypes.npz")
    base, util = csr_from(g, "counts"), csr_from(g, "util")
    cases = {"int16_log": (np.int16, "log", "float32", 2, 10), "int16_linear": (np.int16, "linear", "float32", 2, 2),
             "f32_model64": (np.float32, "log", "float64", 2, 10), "f32_model64_c1": (np.float32, "log", "float64", 1, 10)}
    for name, (cdt, mode, mdt, cores, alpha) in cases.items():
        counts = sp.csr_matrix((np.rint(base.data).astype(cdt), base.indices, base.indptr), shape=base.shape)
        last, hist, users, items = orc.train(40, 80, 5, 0.1, util, 2, util, count_mat=counts, weighted=True, bias=False,
                                             stopping_rounds=5, cores=cores, pre_process_count=mode, alpha=alpha, dtype=mdt)
        assert last == int(g[f"last_iter_{name}"]), name
        assert users.dtype == g[f"users_{name}"].dtype and items.dtype == g[f"items_{name}"].dtype, (name, users.dtype)
        np.testing.assert_allclose(users, g[f"users_{name}"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(items, g[f"items_{name}"], rtol=RTOL, atol=ATOL)
