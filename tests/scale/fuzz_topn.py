"""Randomised sweep (not a pytest) of the ranking surface: eval_topn (seeded, against the oracle's restatement of base_model.py's
compute_hit loop) and rank (against NumPy on the same factors) for random shapes, k, biases, topn lists, candidate counts, users
without test entries and duplicated candidates.  Usage: python tests/scale/fuzz_topn.py [cases] [seed]"""
import sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
from oracle import wmf_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.perf_counter()
for case in range(cases):
    n, m = int(rng.integers(1, 300)), int(rng.integers(2, 5000))
    k = int(rng.choice([1, 2, 8, 16, 33, 64, 128, 256]))
    bias = bool(rng.integers(2))
    topn = np.unique(rng.integers(1, 40, int(rng.integers(1, 5)))).astype(np.int64)
    rand_sampled = int(rng.integers(2 * topn.max() + 2, 2 * topn.max() + 1500))
    seed = int(rng.integers(1 << 30))
    f = k + int(bias)
    model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
    model.users = rng.standard_normal((n, f)).astype(np.float32)
    model.items = rng.standard_normal((m, f)).astype(np.float32)
    test = sp.random(n, m, density=min(1.0, rng.uniform(0.2, 3.0) / m), format="csr", random_state=seed).astype(np.float32)
    test.data[:] = 1.0
    print(f"case {case:3d}: {n} x {m} k={k} bias={int(bias)} topn={topn.tolist()} sampled={rand_sampled} test nnz={test.nnz}", end=" ", flush=True)
    if test.nnz:
        got = model.eval_topn(test_mat=test.copy(), topn=topn, rand_sampled=rand_sampled, random_state=seed)
        want = orc.eval_topn(model.users, model.items, test, topn, rand_sampled, seed, bias)
        for key in want:
            # a hit decided by two scores closer than float32 rounding may flip: allow one entry per list
            assert abs(got[key] - want[key]) <= 1.0 / test.nnz + 1e-12, (key, got[key], want[key])
    cand = rng.integers(0, m, int(rng.integers(1, min(m, 3000) + 1)))          # with repetitions
    user = int(rng.integers(n))
    for tn in (None, 1, int(topn.max())):
        ids = model.rank(cand, user, topn=tn)
        sc = orc.predict(model.users, model.items, np.full(len(cand), user), cand, bias)
        want_len = len(cand) if tn is None else min(tn, len(cand))
        assert len(ids) == want_len, (len(ids), want_len)
        lookup = {}
        for c, s in zip(cand.tolist(), sc.tolist()): lookup[c] = s
        np.testing.assert_allclose([lookup[i] for i in ids], np.sort(sc)[::-1][:want_len], rtol=1e-4, atol=1e-5)
    print("ok")
print(f"{cases} cases in {time.perf_counter() - t0:.1f} s")
