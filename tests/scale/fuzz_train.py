"""Randomised sweep of the CLASS SURFACE (not a pytest): WMF(...).train(...) against the oracle's train() -- random shapes, k, biases,
weighted / unweighted, cores 1 / 2, float32 / float64 / integer counts, 'log' / 'linear', alpha, beta, preprocess_mat, early stopping
-- then eval_prec (three metrics), predict and rank on the trained model.  Usage: python tests/scale/fuzz_train.py [cases] [seed]"""
import sys, time, io, contextlib
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
from oracle import wmf_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
t0 = time.perf_counter()
for case in range(cases):
    k = int(rng.choice([1, 3, 8, 16, 17, 31, 32, 48, 64, 65, 96, 128]))
    # both sides with enough rows for a decent Gramian: with fewer rows than features G = Y^T Y + 0.1 I is ill conditioned and the
    # reference's float32 Gramian product (BLAS order) differs from any other evaluation by 1e-4 in the factors -- not a parity question
    n, m = int(rng.integers(2 * k + 30, 2 * k + 400)), int(rng.integers(2 * k + 30, 2 * k + 300))
    bias = bool(rng.integers(2))
    weighted = bool(rng.integers(5))                       # one in five: the unweighted branch (wmf_model.py:158-162)
    if not weighted: bias = False                          # (the reference's un-weighted branch cannot take a bias model: a ValueError there and here)
    cores = int(rng.choice([1, 2]))
    cdt = rng.choice(["float32", "float64", "int64"])
    mode = rng.choice(["log", "linear"])
    alpha, beta = float(rng.choice([1.0, 10.0, 40.0])), float(rng.choice([0.5, 1.0, 2.0]))
    iters, stop_r = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    min_imp = float(rng.choice([1e-4, 0.05]))
    pre_mat = bool(rng.integers(2))
    dens = rng.uniform(0.02, 0.3)
    counts = sp.random(n, m, density=dens, format="csr", random_state=int(rng.integers(1 << 30)),
                       data_rvs=lambda s: rng.integers(1, 9, s)).astype(cdt)
    util = counts.copy().astype(np.float32); util.data[:] = 1.0
    ev = util if rng.integers(2) else sp.random(n, m, density=0.05, format="csr", random_state=int(rng.integers(1 << 30))).astype(np.float32)
    if ev.nnz == 0: ev = util
    kw = dict(utility_mat=util, iterations=iters, eval_mat=ev, count_mat=counts, alpha=alpha, cores=cores, stopping_rounds=stop_r,
              min_improvement=min_imp, pre_process_count=mode, beta=beta, preprocess_mat=pre_mat)
    tagline = (f"case {case:3d}: {n}x{m} k={k} bias={int(bias)} weighted={int(weighted)} cores={cores} counts={cdt} {mode} a={alpha} b={beta} "
               f"it={iters} stop={stop_r} pre={int(pre_mat)}")
    print(tagline, end=" ", flush=True)
    model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=weighted, bias=bias)
    with contextlib.redirect_stdout(io.StringIO()):
        last = model.train(**kw)
    okw = dict(kw); okw.pop("utility_mat"); okw.pop("iterations"); okw.pop("eval_mat")
    want_last, hist, wu, wi = orc.train(m, n, k, 0.1, util, iters, ev, weighted=weighted, bias=bias, **okw)
    f64 = wu.dtype == np.float64
    # (float64 path: the reference's FIRST Gramian is a float32 product of the float32 initial items -- BLAS order, ~1e-6 -- and
    # the device forms it in float64: the goldens' 2e-5, tests/test_gpu_parity.py)
    tol = 1e-4 if f64 else (1.2e-3 if (mode == 'linear' and alpha >= 40) else 3e-4)      # (weights up to 320: two float32 evaluations of cond ~ 1e4 systems)
    eu = np.linalg.norm(model.users - wu) / np.linalg.norm(wu)
    ei = np.linalg.norm(model.items - wi) / np.linalg.norm(wi)
    assert model.users.dtype == wu.dtype and model.items.dtype == wi.dtype, (model.users.dtype, wu.dtype)
    # early stopping compares consecutive MSEs with a relative margin: a last-bit difference may flip it only when the oracle's own margin is tiny
    margins = [abs(hist[i] * (1 + min_imp) - (hist[i - 1] if i else -np.inf)) / abs(hist[i]) for i in range(len(hist))]
    if last != want_last:
        assert min(margins) < 1e-4, (last, want_last, hist)
        print(f"(stop point differs at a margin of {min(margins):.1e}: skipped)")
        continue
    for metric in ("mse", "rmse", "mae"):
        a, b = model.eval_prec(ev, metric), orc.eval_prec(wu, wi, ev, bias, metric)
        assert abs(a - b) <= (2e-4 if f64 else 2e-3) * abs(b) + 1e-12, (metric, a, b)      # (the device evaluates in float32 either way)
    us, its = rng.integers(0, n, 7), rng.integers(0, m, 7)
    np.testing.assert_allclose(model.predict(us, its), orc.predict(wu, wi, us, its, bias), rtol=(2e-3 if f64 else 5e-3), atol=(2e-4 if f64 else 5e-4))
    cand = rng.choice(m, min(m, 25), replace=False)
    got = model.rank(cand, int(us[0]), topn=5)
    sc = orc.predict(model.users, model.items, np.full(len(cand), us[0]), cand, bias)      # scores on the model's OWN factors: ties aside, the order is determined
    top = np.sort(sc)[::-1][:5]
    np.testing.assert_allclose(np.sort(sc[[list(cand).index(i) for i in got]])[::-1], top, rtol=1e-5, atol=1e-6)
    worst = max(worst, eu / tol, ei / tol)
    print(f"last={last} users {eu:.1e} items {ei:.1e}" + ("" if max(eu, ei) <= tol else "   <-- ABOVE TOLERANCE"))
print(f"{cases} cases in {time.perf_counter() - t0:.1f} s; worst error / tolerance = {worst:.2f}")
sys.exit(0 if worst <= 1.0 else 1)
