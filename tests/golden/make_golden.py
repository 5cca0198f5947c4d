#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the *reference* WMF class.

Run in the build container only (the reference lives at /root/reference and never travels):

    python3 -B tests/golden/make_golden.py

The reference package cannot be imported whole (its ``__init__`` needs compiled Cython
modules), so the two files on the WMF path are loaded in isolation:
an empty ``RecModel`` package object bypasses ``__init__``, an empty ``sharedmem`` module
satisfies the unused import at base_model.py:8, and the MKL runtime that ``MKLThreads``
dlopens (base_model.py:191) is pointed at /opt/conda/lib.  Nothing of the reference is
copied: the outputs are data (inputs + expected outputs) saved as ``.npz``.
"""
import ctypes
import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

sys.dont_write_bytecode = True
REF = "/root/reference/RecModel"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.modules.setdefault("sharedmem", types.ModuleType("sharedmem"))
    pkg = types.ModuleType("RecModel")
    pkg.__path__ = [REF]
    sys.modules["RecModel"] = pkg
    mods = {}
    for name in ("base_model", "wmf_model"):
        spec = importlib.util.spec_from_file_location(f"RecModel.{name}", f"{REF}/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"RecModel.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    mods["base_model"].MKLThreads._mkl_rt = ctypes.CDLL("/opt/conda/lib/libmkl_rt.so")
    return mods["wmf_model"].WMF


def make_counts(n_users, n_items, mean_deg, seed, dtype):
    """Small ragged count matrix with the edge cases SURVEY.md section 8c asks for:
    an empty row, an empty column, a stored zero, one very heavy row and one heavy column."""
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = np.maximum(rng.poisson(mean_deg, n_users), 1)
    deg[3] = 0                       # empty user row
    deg[7] = n_items - 5             # very heavy row
    pop = 1.0 / np.arange(1, n_items + 1) ** 0.6
    pop[11] = 0.0                    # item never chosen -> empty row of the transpose
    pop /= pop.sum()
    rows, cols = [], []
    for u in range(n_users):
        c = np.sort(rng.choice(n_items, size=min(deg[u], n_items - 1), replace=False, p=pop))
        rows.append(np.full(len(c), u))
        cols.append(c)
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = (1 + rng.geometric(0.5, len(rows))).astype(dtype)
    vals[5] = 0                      # a *stored* zero: still contributes (w+1)*y to b
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n_users, n_items), dtype=dtype)
    assert m.nnz == len(vals), "stored zero must stay stored"
    return m


def csr_fields(prefix, m):
    return {f"{prefix}_indptr": m.indptr.copy(), f"{prefix}_indices": m.indices.copy(),
            f"{prefix}_data": m.data.copy(), f"{prefix}_shape": np.array(m.shape)}


def main():
    WMF = load_reference()
    n_users, n_items = 300, 120

    # ---- a1: initial item factors ------------------------------------------------
    init = {}
    for bias in (False, True):
        mdl = WMF(num_items=37, num_users=5, dim=6, gamma=0.1, weighted=True, bias=bias, seed=1993)
        init[f"items_bias{int(bias)}"] = mdl.items.copy()
    mdl = WMF(num_items=9, num_users=5, dim=4, gamma=0.1, weighted=True, seed=7)
    init["items_seed7"] = mdl.items.copy()
    np.savez_compressed(f"{OUT}/init.npz", **init)

    # ---- a3/a4: half steps from the fixed init, bias x count dtype ---------------
    for bias in (False, True):
        for cdt in ("float32", "float64"):
            counts = make_counts(n_users, n_items, 6, seed=11, dtype=cdt)
            mdl = WMF(num_items=n_items, num_users=n_users, dim=16, gamma=0.1, weighted=True,
                      bias=bias, seed=1993)
            C = counts.copy()
            C.data = 10 * np.log(1 + C.data)          # the train() defaults, wmf_model.py:120
            CT = C.T.tocsr()
            items0 = mdl.items.copy()
            if bias:
                users1 = mdl.recompute_factors_bias(items0.copy(), C.copy(), 0.1, cores=1)
                items1 = mdl.recompute_factors_bias(users1.copy(), CT.copy(), 0.1, cores=1)
                users2 = mdl.recompute_factors_bias(items1.copy(), C.copy(), 0.1, cores=1)
            else:
                users1 = mdl.recompute_factors(items0, C, 0.1)
                items1 = mdl.recompute_factors(users1, CT, 0.1)
                users2 = mdl.recompute_factors(items1, C, 0.1)
            out = dict(items0=items0, users1=users1, items1=items1, users2=users2, gamma=0.1)
            out.update(csr_fields("C", C))
            out.update(csr_fields("CT", CT))
            out.update(csr_fields("counts", counts))
            np.savez_compressed(f"{OUT}/half_bias{int(bias)}_{cdt}.npz", **out)

    # ---- a5/a6: the Pool variants return float64 -----------------
    counts = make_counts(40, 30, 4, seed=5, dtype="float64")
    C = counts.copy()
    C.data = 10 * np.log(1 + C.data)
    out = {}
    for bias in (False, True):
        mdl = WMF(num_items=30, num_users=40, dim=8, gamma=0.1, weighted=True, bias=bias, seed=1993)
        if bias:
            u = mdl.recompute_factors_bias_par(mdl.items.copy(), C.copy(), 0.1, cores=2)
        else:
            u = mdl.recompute_factors_par(mdl.items, C, 0.1, cores=2)
        out[f"users_par_bias{int(bias)}"] = u
        out[f"items0_bias{int(bias)}"] = mdl.items.copy()
    out.update(csr_fields("C", C))
    np.savez_compressed(f"{OUT}/half_par.npz", **out)

    # ---- a2/a7/a8/a10: full train() control flow, predict, eval_prec, rank -------
    for tag, bias, iters, rounds, min_imp in (("run", False, 4, 3, 1e-4), ("stop", False, 8, 2, 0.5),
                                              ("bias", True, 3, 3, 1e-4)):
        counts = make_counts(n_users, n_items, 6, seed=23, dtype="float64")
        util = counts.copy()
        util.data = np.minimum(util.data, 5.0)
        mdl = WMF(num_items=n_items, num_users=n_users, dim=12, gamma=0.1, weighted=True, bias=bias,
                  seed=1993)
        history = []
        inner = mdl.eval_prec

        def recorder(mat, metric="mse", _inner=inner, _h=history):
            v = _inner(mat, metric)
            _h.append(v)
            return v
        mdl.eval_prec = recorder
        devnull = open(os.devnull, "w")
        stdout, sys.stdout = sys.stdout, devnull      # bias+cores==1 prints per iteration (:153)
        try:
            last = mdl.train(utility_mat=util, iterations=iters, verbose=0, eval_mat=util,
                             count_mat=counts, cores=1, stopping_rounds=rounds, min_improvement=min_imp)
        finally:
            sys.stdout = stdout
        del mdl.eval_prec
        pu = np.array([0, 1, 2, 7, 50, 299, 3])
        pi = np.array([0, 5, 11, 119, 60, 1, 2])
        cand = np.arange(0, n_items, 2)
        out = dict(last_iter=last, mse=np.array(history), users=mdl.users, items=mdl.items,
                   pred_users=pu, pred_items=pi, pred=mdl.predict(pu, pi),
                   pred_one_user=mdl.predict(np.array([7]), cand),
                   rank_cand=cand, rank_top5=mdl.rank(cand, 7, topn=5),
                   rank_top40=mdl.rank(cand, 7, topn=40), rank_all=mdl.rank(cand, 7),
                   rank_list0=np.array(mdl.rank(cand, [1, 2], topn=3)),
                   mse_final=mdl.eval_prec(util), rmse_final=mdl.eval_prec(util, "rmse"),
                   mae_final=mdl.eval_prec(util, "mae"),
                   iterations=iters, stopping_rounds=rounds, min_improvement=min_imp, dim=12)
        out.update(csr_fields("counts", counts))
        out.update(csr_fields("util", util))
        np.savez_compressed(f"{OUT}/train_{tag}.npz", **out)

    # ---- un-weighted branch (weighted=None is the constructor default) -----------
    counts = make_counts(n_users, n_items, 6, seed=31, dtype="float64")
    util = counts.copy()
    util.data = np.minimum(util.data, 5.0)
    mdl = WMF(num_items=n_items, num_users=n_users, dim=10, gamma=0.5, seed=1993)
    last = mdl.train(utility_mat=util, iterations=2, eval_mat=util, stopping_rounds=5)
    out = dict(last_iter=last, users=mdl.users, items=mdl.items, mse_final=mdl.eval_prec(util))
    out.update(csr_fields("util", util))
    np.savez_compressed(f"{OUT}/train_unweighted.npz", **out)

    # ---- sampled Recall@N (eval_topn -> compute_hit -> rank), seeded ---------------
    for bias in (False, True):
        counts = make_counts(80, 120, 6, seed=41 + bias, dtype="float64")
        mdl = WMF(num_items=120, num_users=80, dim=8, gamma=0.1, weighted=True, bias=bias, seed=1993)
        mdl.train(utility_mat=counts.copy(), count_mat=counts.copy(), iterations=2, eval_mat=counts.copy(), cores=1,
                  stopping_rounds=5)
        rng = np.random.Generator(np.random.PCG64(5 + bias))
        coo = counts.tocoo()
        keep = rng.random(coo.nnz) < 0.15                       # a sparse test split; several users end up without test items
        keep &= coo.row % 7 != 3
        test = sp.csr_matrix((coo.data[keep], (coo.row[keep], coo.col[keep])), shape=counts.shape)
        topn = np.array([1, 5, 10])
        res = mdl.eval_topn(test_mat=test.copy(), topn=topn, rand_sampled=40, random_state=11)
        out = dict(users=mdl.users, items=mdl.items, topn=topn, rand_sampled=40, random_state=11,
                   recall=np.array([res[f"Recall@{n}"] for n in topn], dtype=np.float64), n_test=len(test.nonzero()[0]))
        out.update(csr_fields("test", test))
        np.savez_compressed(f"{OUT}/eval_topn_bias{int(bias)}.npz", **out)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(f"{OUT}/{f}"))


def make_train_par(WMF):
    """a5 / a6 inside train(): cores = 2 on a float64 count matrix -- the Pool variants leave float64 factors, so every
    half step after the first runs on float64 inputs (wmf_model.py:147-157, :242-265)."""
    counts = make_counts(120, 60, 5, seed=51, dtype="float64")
    util = counts.copy()
    util.data = np.minimum(util.data, 5.0)
    out = {}
    for bias in (False, True):
        mdl = WMF(num_items=60, num_users=120, dim=8, gamma=0.1, weighted=True, bias=bias, seed=1993)
        # (no per-iteration recorder on the instance here: Pool pickles the bound row function and the model with it)
        last = mdl.train(utility_mat=util, iterations=3, verbose=0, eval_mat=util, count_mat=counts, cores=2,
                         stopping_rounds=5)
        assert mdl.users.dtype == np.float64 and mdl.items.dtype == np.float64
        out[f"last_iter_bias{int(bias)}"] = last
        out[f"mse_final_bias{int(bias)}"] = mdl.eval_prec(util)
        out[f"users_bias{int(bias)}"] = mdl.users
        out[f"items_bias{int(bias)}"] = mdl.items
    out.update(csr_fields("counts", counts))
    out.update(csr_fields("util", util))
    np.savez_compressed(f"{OUT}/train_par.npz", **out)
    print("train_par.npz", os.path.getsize(f"{OUT}/train_par.npz"))


def make_train_par_int(WMF):
    """cores = 2 on an INTEGER count matrix (round 3): `alpha * np.log(1 + beta * data)` of int64 data is float64
    (wmf_model.py:120), and with 'linear' the int64 weights promote the row products to float64 (:123, :285-287) -- either
    way the Pool variants return float64 rows and training continues on float64 factors."""
    counts = make_counts(90, 50, 5, seed=61, dtype="float64")
    counts = sp.csr_matrix((np.rint(counts.data).astype(np.int64), counts.indices, counts.indptr), shape=counts.shape)
    util = counts.astype(np.float64)
    util.data = np.minimum(util.data, 5.0)
    out = {}
    for mode in ("log", "linear"):
        mdl = WMF(num_items=50, num_users=90, dim=6, gamma=0.1, weighted=True, bias=(mode == "linear"), seed=1993)
        last = mdl.train(utility_mat=util, iterations=2, verbose=0, eval_mat=util, count_mat=counts, cores=2,
                         stopping_rounds=5, pre_process_count=mode, alpha=(10 if mode == "log" else 2))
        assert mdl.users.dtype == np.float64 and mdl.items.dtype == np.float64
        out[f"last_iter_{mode}"] = last
        out[f"mse_final_{mode}"] = mdl.eval_prec(util)
        out[f"users_{mode}"] = mdl.users
        out[f"items_{mode}"] = mdl.items
    out.update(csr_fields("counts", counts))
    out.update(csr_fields("util", util))
    np.savez_compressed(f"{OUT}/train_par_int.npz", **out)
    print("train_par_int.npz", os.path.getsize(f"{OUT}/train_par_int.npz"))


def make_train_dtypes(WMF):
    """Which dtype the factors end in follows NumPy's promotion of the MODEL dtype and the TRANSFORMED counts (round 4):
    int16 counts with cores = 2 stay float32 ('log': np.log of int16 is float32; 'linear': float32 * int16 is float32), a
    float64 model dtype with float32 counts gives float64 rows (wmf_model.py:119-123, :237-239, :285-287)."""
    base = make_counts(80, 40, 5, seed=71, dtype="float64")
    util = base.copy()
    util.data = np.minimum(util.data, 5.0)
    out = {}
    cases = {"int16_log": dict(cdt=np.int16, mode="log", mdt="float32", cores=2, alpha=10),
             "int16_linear": dict(cdt=np.int16, mode="linear", mdt="float32", cores=2, alpha=2),
             "f32_model64": dict(cdt=np.float32, mode="log", mdt="float64", cores=2, alpha=10),
             "f32_model64_c1": dict(cdt=np.float32, mode="log", mdt="float64", cores=1, alpha=10)}
    for name, c in cases.items():
        counts = sp.csr_matrix((np.rint(base.data).astype(c["cdt"]), base.indices, base.indptr), shape=base.shape)
        mdl = WMF(num_items=40, num_users=80, dim=5, gamma=0.1, weighted=True, bias=False, seed=1993, dtype=c["mdt"])
        last = mdl.train(utility_mat=util, iterations=2, verbose=0, eval_mat=util, count_mat=counts, cores=c["cores"],
                         stopping_rounds=5, pre_process_count=c["mode"], alpha=c["alpha"])
        out[f"last_iter_{name}"] = last
        out[f"users_{name}"] = mdl.users
        out[f"items_{name}"] = mdl.items
        out[f"mse_final_{name}"] = mdl.eval_prec(util)
        print(name, mdl.users.dtype, mdl.items.dtype)
    out.update(csr_fields("counts", base))
    out.update(csr_fields("util", util))
    np.savez_compressed(f"{OUT}/train_dtypes.npz", **out)
    print("train_dtypes.npz", os.path.getsize(f"{OUT}/train_dtypes.npz"))


if __name__ == "__main__":
    if sys.argv[1:] == ["train_dtypes"]:                 # only the file added in round 4
        make_train_dtypes(load_reference())
        sys.exit(0)
    if sys.argv[1:] == ["train_par_int"]:                # only the file added in round 3
        make_train_par_int(load_reference())
        sys.exit(0)
    if sys.argv[1:] == ["train_par"]:                    # only the file added in round 2 (the others stay byte-identical)
        make_train_par(load_reference())
        sys.exit(0)
    main()
