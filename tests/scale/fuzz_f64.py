"""Randomised sweep of the float64 half step (wmf_recompute_factors_f64_host: Gramian, low-rank rows, direct rows, LU fallback)
against the NumPy oracle in float64: random widths (every block count per thread, wave and workgroup teams), biases with and
without negative weights, degree laws on both sides of the low-rank switch (a quarter of the rows with 1 .. 32 entries).
Usage: python tests/scale/fuzz_f64.py [cases] [seed]"""
import sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
from oracle import wmf_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
FS = [1, 2, 3, 4, 5, 8, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 68, 69, 84, 85, 100, 120, 121, 129, 148, 149, 192, 193, 200, 256, 257, 260]
worst = 0.0
t0 = time.perf_counter()
for case in range(cases):
    f = int(rng.choice(FS))
    bias = bool(rng.integers(2)) and f >= 2
    k = f - int(bias)
    m = int(rng.integers(max(2 * f, 40), 3 * f + 150))
    n = int(rng.integers(10, 150 if f > 130 else 300))
    law = rng.choice(["short", "long", "mixed", "tiny"])
    rows, cols = [], []
    for u in range(n):
        if law == "short": d = rng.poisson(10)
        elif law == "long": d = int(rng.integers(33, min(m, 300)))
        elif law == "tiny": d = int(rng.integers(0, 4))
        else: d = int(rng.choice([0, 1, 4, 5, 31, 32, 33, 34, 64, min(m, 200)]))
        d = min(d, m)
        c = rng.choice(m, d, replace=False)
        rows += [u] * d; cols += c.tolist()
    vals = 10 * np.log(1 + rng.integers(1, 8, len(rows))).astype(np.float64)
    if len(vals) and rng.integers(3) == 0:
        vals[rng.integers(len(vals))] = 0.0                         # a stored zero
    C = sp.csr_matrix((vals, (rows, cols)), shape=(n, m))
    model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias, seed=int(rng.integers(1 << 30)))
    Y = model.items.astype(np.float64)
    neg = bias and bool(rng.integers(2))
    if bias:
        Y[:, 0] = np.linspace(-5, 30, m) if neg else 0.5 * Y[:, 0]   # neg: about a third of the weights go negative -> LU rows
    fn = model.recompute_factors_bias_par if bias else model.recompute_factors_par
    got = fn(Y, C, 0.1)
    want = (orc.recompute_factors_bias if bias else orc.recompute_factors)(Y, C, 0.1, dtype="float64")
    ok = np.linalg.norm(want, axis=1) < 1e3 if neg else np.ones(n, dtype=bool)
    err = np.linalg.norm(got[ok] - want[ok]) / max(np.linalg.norm(want[ok]), 1e-30)
    tol = 1e-8 if neg else 1e-10
    flag = "" if err <= tol and np.isfinite(got).all() else "   <-- ABOVE TOLERANCE"
    worst = max(worst, err / tol if np.isfinite(err) else 1e9)
    empty = np.diff(C.indptr) == 0
    assert not got[empty].any(), "rows without entries must be exactly zero"
    print(f"case {case:3d}: f={f:3d} bias={int(bias)} neg={int(neg)} n={n:3d} m={m:4d} law={law:6s} nnz={C.nnz:6d} rel.err {err:.2e}{flag}")
print(f"{cases} cases in {time.perf_counter() - t0:.1f} s; worst error / tolerance = {worst:.2f}")
sys.exit(0 if worst <= 1.0 else 1)
