"""Diagnostic (GPU, hand-run): per-row errors of one generated case of the fuzz_parity.py family.
Usage: python tests/scale/diag_fuzz_case.py f bias n m law seed"""
import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
from oracle import c_oracle

f, bias, n, m, law, seed = int(sys.argv[1]), bool(int(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
rng = np.random.default_rng(seed)
k = f - int(bias)
rows, cols = [], []
for u in range(n):
    if law == "poisson": d = rng.poisson(12)
    elif law == "heavy": d = int(rng.integers(30, min(m, 700)))
    elif law == "tiny": d = int(rng.integers(0, 4))
    else: d = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 100, min(m, 300)]))
    d = min(d, m)
    c = rng.choice(m, d, replace=False)
    rows += [u] * d; cols += c.tolist()
vals = (10 * np.log(1 + rng.integers(1, 8, len(rows)))).astype(np.float32)
C = sp.csr_matrix((vals, (rows, cols)), shape=(n, m))
model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias, seed=int(rng.integers(1 << 30)))
Y = model.items
if bias:
    Y = Y.copy(); Y[:, 0] *= 0.5
want = c_oracle.half_step(Y, sp.csr_matrix((C.data.astype(np.float64), C.indices, C.indptr), shape=C.shape), 0.1, bias)
got = (model.recompute_factors_bias if bias else model.recompute_factors)(Y, C, 0.1).astype(np.float64)
e = np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
deg = np.diff(C.indptr)
print("overall", np.linalg.norm(got - want) / np.linalg.norm(want))
order = np.argsort(-e)
for r in order[:15]:
    print(f"row {r:4d} d={deg[r]:4d} err {e[r]:.2e}")
print("rows above 1e-3:", int((e > 1e-3).sum()), "of", n, "; degrees of those:", sorted(deg[e > 1e-3].tolist())[:20])
got2 = (model.recompute_factors_bias if bias else model.recompute_factors)(Y, C, 0.1).astype(np.float64)
diff = np.abs(got2 - got).max(axis=1)
print("run-to-run: rows that differ", int((diff > 0).sum()), "max abs diff", float(diff.max()))
e2 = np.linalg.norm(got2 - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
print("second run rows above 1e-3:", int((e2 > 1e-3).sum()), "; same set:", bool(((e > 1e-3) == (e2 > 1e-3)).all()))
import os
if os.environ.get("DIAG_FLAGS"):
    from recmodel_amd import _lib
    lib = _lib.load()
    lib.wmf_debug_set_flags(int(os.environ["DIAG_FLAGS"]))
    got3 = (model.recompute_factors_bias if bias else model.recompute_factors)(Y, C, 0.1).astype(np.float64)
    got4 = (model.recompute_factors_bias if bias else model.recompute_factors)(Y, C, 0.1).astype(np.float64)
    print("with debug flags: run-to-run rows that differ", int((np.abs(got4 - got3).max(axis=1) > 0).sum()))
    lib.wmf_debug_set_flags(0)
    e3 = np.linalg.norm(got3 - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
    print("with debug flags", os.environ["DIAG_FLAGS"], ": rows above 1e-3:", int((e3 > 1e-3).sum()))
