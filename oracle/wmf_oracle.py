"""CPU oracle for the WMF/ALS hot path -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the arithmetic of the reference
``RecModel/wmf_model.py`` + ``RecModel/base_model.py`` (titoeb/RecModel), written
from the specification in SURVEY.md section 8(a).  It is the *checker* for the HIP
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it.  Nothing under ``recmodel_amd/`` imports it,
and the product path raises if the HIP library is missing instead of falling
back to this code.

Parity status: PINNED.  ``tests/golden/*.npz`` were produced by importing the
reference ``WMF`` class itself in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` checks every
function below against those vectors.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).
"""
import numpy as np
import scipy.sparse as sp


# --------------------------------------------------------------------------- a1
def init_items(num_items, dim, bias=False, seed=1993, dtype="float32"):
    """Initial item factors.  RecModel/wmf_model.py:10-17.

    The reference seeds the *global* legacy NumPy RNG and draws
    ``random((num_items, dim [+1]))`` as float64, then casts to ``dtype``.
    """
    np.random.seed(seed)
    width = dim + 1 if bias else dim
    return np.random.random((num_items, width)).astype(dtype)


# --------------------------------------------------------------------------- a2
def confidence_transform(values, alpha=10, beta=1, mode="log"):
    """Count -> confidence weight.  RecModel/wmf_model.py:119-126."""
    if mode == "log":
        return alpha * np.log(1 + beta * values)
    if mode == "linear":
        return alpha * values
    raise ValueError(f"Pre_process_count {mode} is not implement please use log or linear.")


# --------------------------------------------------------------------------- a3
def gramian(Y, lam, dtype="float32"):
    """``Y^T Y + lam * I`` in the model dtype.  RecModel/wmf_model.py:215."""
    return np.dot(Y.T, Y) + lam * np.eye(Y.shape[1], dtype=dtype)


def solve_row(G, Y, idx, w):
    """One row of the ALS update.  RecModel/wmf_model.py:231-239.

    ``A = G + U^T diag(w) U``, ``b = (w + 1)^T U`` with ``U = Y[idx]``; LU solve.
    Every *stored* entry contributes (stored zeros too); duplicates are not merged.
    The product ``U * w[:, None]`` promotes to float64 when ``w`` is float64, which
    is what makes the reference compute in double for double-precision counts.
    """
    U = Y[idx, :]
    A = np.dot(U.T, U * w[:, np.newaxis]) + G
    b = np.dot(w + 1, U)
    return np.linalg.solve(A, b)


def solve_row_in_slabs(G, slabs):
    """``solve_row`` for a row too long to gather at once (a power-law head with millions of stored entries): the same
    two sums of RecModel/wmf_model.py:237-239, ``A = G + sum_s U_s^T diag(w_s) U_s`` and ``b = sum_s (w_s + 1)^T U_s``,
    taken over consecutive slabs ``(U_s, w_s)`` of the row's gathered factors, then the same LU solve."""
    A = np.array(G, dtype=np.float64, copy=True)
    b = np.zeros(A.shape[0], dtype=np.float64)
    for U, w in slabs:
        A += np.dot(U.T, U * w[:, np.newaxis])
        b += np.dot(w + 1, U)
    return np.linalg.solve(A, b)


def recompute_factors(Y, C, lam, dtype="float32", out_dtype=None):
    """No-bias half step.  RecModel/wmf_model.py:213-240.

    Y: [m, k] fixed factors; C: CSR [n, m] of confidence weights.  Returns [n, k] in
    ``dtype``; rows without stored entries are zero (``:223-225``).  ``out_dtype`` models the
    ``cores>1`` variant (``:242-250,289-309``): same Gramian in the model dtype, but the per-row
    float64 solutions are stacked without the cast back to ``dtype``.
    """
    G = gramian(Y, lam, dtype)
    n = C.shape[0]
    out = np.empty((n, Y.shape[1]), dtype=out_dtype or dtype)
    indptr, indices, data = C.indptr, C.indices, C.data
    for r in range(n):
        lo, hi = indptr[r], indptr[r + 1]
        if hi == lo:
            out[r, :] = 0
        else:
            out[r, :] = solve_row(G, Y, indices[lo:hi], data[lo:hi])
    return out


# --------------------------------------------------------------------------- a4
def recompute_factors_bias(Y, C, lam, dtype="float32", out_dtype=None):
    """Half step with biases.  RecModel/wmf_model.py:311-351.

    Column 0 of ``Y`` is the fixed side's bias: it is saved, the column is set to 1
    (the caller passes a copy, ``:151-152``; we copy here), the Gramian is taken of the
    modified matrix (``:328-332``) and each stored weight becomes ``w - bias[idx]``
    (``:343``).  Output column 0 is the updated side's bias.  There is no empty-row
    branch in the reference; ``solve(G, 0)`` yields zeros.
    """
    Y = Y.copy()
    bias = Y[:, 0].copy()
    Y[:, 0] = 1
    G = gramian(Y, lam, dtype)
    n = C.shape[0]
    out = np.empty((n, Y.shape[1]), dtype=out_dtype or dtype)
    indptr, indices, data = C.indptr, C.indices, C.data
    for r in range(n):
        lo, hi = indptr[r], indptr[r + 1]
        idx = indices[lo:hi]
        out[r, :] = solve_row(G, Y, idx, data[lo:hi] - bias[idx])
    return out


# --------------------------------------------------------------------------- a7
def predict(users_f, items_f, users, items, bias=False):
    """Scores for (user, item) pairs.  RecModel/wmf_model.py:191-211."""
    if isinstance(users, (list, np.ndarray)) and isinstance(items, (list, np.ndarray)):
        if len(users) != len(items) and not (len(users) == 1 or len(items) == 0):
            raise ValueError("users and items need to have the same length or only one user / item needs to be provided.")
    if not bias:
        return (users_f[users, :] * items_f[items, :]).sum(axis=1)
    return ((users_f[:, 1:][users, :] * items_f[:, 1:][items, :]).sum(axis=1)
            + users_f[:, 0][users] + items_f[:, 0][items])


# --------------------------------------------------------------------------- a8
def eval_prec(users_f, items_f, utility_mat, bias=False, metric="mse"):
    """MSE / RMSE / MAE over the stored-nonzero entries.  RecModel/base_model.py:150-179."""
    metric = metric.upper()
    if metric not in ("MSE", "RMSE", "MAE"):
        raise ValueError("Metric {metric} is not implemented.")
    rows, cols = utility_mat.nonzero()
    pred = predict(users_f, items_f, rows, cols, bias).reshape(1, -1)
    diff = np.asarray(utility_mat[rows, cols]) - pred
    if metric == "RMSE":
        return np.sqrt(np.mean(np.square(diff)))
    if metric == "MSE":
        return np.mean(np.square(diff))
    return np.mean(np.abs(diff))


# -------------------------------------------------------------------------- a10
def rank(users_f, items_f, items, user, topn=None, bias=False):
    """Top-n candidate items for one user, best first.  RecModel/wmf_model.py:25-47."""
    items = np.asarray(items)
    if topn is None:
        topn = len(items)
    scores = predict(users_f, items_f, user, items, bias)
    if len(scores) * 0.5 > topn:
        part = np.argpartition(scores, list(range(-topn, 0, 1)))[-topn:]
        return items[part][::-1]
    return items[np.argsort(scores)[-topn:]][::-1]


def eval_topn(users_f, items_f, test_mat, topn, rand_sampled=1000, random_state=None, bias=False, return_hits=False):
    """Recall@N of each test entry among ``rand_sampled`` random candidates.
    RecModel/base_model.py:100-148 (eval_topn, cores == 1) with compute_hit :51-98 inlined; the legacy
    np.random stream is consumed in the same order: per user WITH test entries one randint draw of
    rand_sampled + 1 candidates, then one randint draw of the slot the test item is written to."""
    if random_state is not None:
        np.random.seed(random_state)
    if not isinstance(topn, np.ndarray):
        raise ValueError("Topn has to be a np.array")
    num_items = items_f.shape[0]
    hits = np.zeros(topn.shape, dtype="float32")
    for user in range(test_mat.shape[0]):
        test_idx = test_mat.indices[test_mat.indptr[user]:test_mat.indptr[user + 1]]
        if len(test_idx) == 0:
            continue
        cand = np.random.randint(0, num_items, size=(rand_sampled + 1))
        slot = np.random.randint(0, rand_sampled - (2 * topn.max()))
        for item in test_idx:
            cand[slot] = item
            best = rank(users_f, items_f, cand, user, topn=topn.max(), bias=bias)
            for pos in range(len(topn)):
                if item in best[:topn[pos]]:
                    hits[pos] += 1
    recall = hits / len(test_mat.nonzero()[0])
    out = {f"Recall@{topn[pos]}": recall[pos] for pos in range(len(topn))}
    return (out, hits) if return_hits else out


# ------------------------------------------------------------- un-weighted branch
def unweighted_half_steps(items_f, utility_mat, gamma, dim, dtype="float32"):
    """One iteration of the closed-form un-weighted branch.  RecModel/wmf_model.py:85,88."""
    eye = gamma * np.eye(dim, dtype=dtype)
    P = np.dot(np.linalg.inv(np.dot(items_f.T, items_f) + eye), items_f.T)
    users_f = sp.csr_matrix.dot(P, utility_mat.T).T.copy()
    Q = np.dot(np.linalg.inv(np.dot(users_f.T, users_f) + eye), users_f.T)
    items_f = sp.csr_matrix.dot(Q, utility_mat).T.copy()
    return users_f, items_f


# --------------------------------------------------------------------------- a2
def train(num_items, num_users, dim, gamma, utility_mat, iterations, eval_mat, count_mat=None,
          weighted=True, bias=False, seed=1993, dtype="float32", alpha=10, stopping_rounds=3,
          min_improvement=0.0001, pre_process_count="log", beta=1, preprocess_mat=False, cores=1):
    """``WMF(...).train(...)`` as a function.  RecModel/wmf_model.py:49-189.

    ``cores > 1`` (``:147-157``): the Pool variants stack their rows without the cast back to ``dtype``
    (``:246``, ``:261``), so with a float64 count matrix the factors are float64 from the first half
    step on and every later Gramian and row system is formed from float64 inputs.

    Returns ``(last_iter, mse_history, users, items)``.
    Early stopping (``:164-168,179-180``): the counter advances whenever
    ``mse * (1 + min_improvement) > previous mse`` (always on the first iteration since
    the previous value starts at ``-inf``) and resets otherwise.
    """
    items_f = init_items(num_items, dim, bias, seed, dtype)
    users_f = None
    utility_mat = utility_mat.copy()
    if preprocess_mat:
        if pre_process_count == "log":
            utility_mat.data = alpha * np.log(1 + beta * utility_mat.data)
        elif pre_process_count == "linear":
            utility_mat.data = alpha * utility_mat.data
    history = []
    last_mse, stall = -np.inf, 0
    it = -1
    if weighted is not True:
        for it in range(iterations):
            users_f, items_f = unweighted_half_steps(items_f, utility_mat, gamma, dim, dtype)
            mse = eval_prec(users_f, items_f, eval_mat, bias)
            history.append(mse)
            stall = stall + 1 if mse * (1 + min_improvement) > last_mse else 0
            last_mse = mse
            if stall >= stopping_rounds:
                break
        return it, history, users_f, items_f

    C = count_mat.copy()
    C.data = confidence_transform(C.data, alpha, beta, pre_process_count)
    CT = C.T.tocsr()
    step = recompute_factors_bias if bias else recompute_factors
    # the Pool variants stack np.linalg.solve results of Y_rel (model dtype) x C.data products: float64 for float64 AND for
    # integer weights (int64 from 'linear' on integer counts promotes the products, :285-287)
    out_dtype = "float64" if cores > 1 and np.result_type(np.dtype(dtype), C.dtype) == np.float64 else None
    for it in range(iterations):
        users_f = step(items_f, C, gamma, dtype, out_dtype)
        items_f = step(users_f, CT, gamma, dtype, out_dtype)
        mse = eval_prec(users_f, items_f, eval_mat, bias)
        history.append(mse)
        stall = stall + 1 if mse * (1 + min_improvement) > last_mse else 0
        last_mse = mse
        if stall >= stopping_rounds:
            break
    return it, history, users_f, items_f
