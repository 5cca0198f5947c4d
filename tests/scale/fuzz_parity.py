"""Randomised parity sweep (not a pytest): many (k, bias, shape, degree law) combinations, one half step each through the
host entry point against the C oracle (oracle/wmf_oracle.c, float64).  Widths are drawn to hit every kernel family and
the boundaries between them (f = 16, 17, 32, 33, 48, 49, 64, 65, 113, 128, 129, 144, 145, 160, 161, 256, 257, 258, 260).
One case in four has 1000 .. 3000 rows so that every resident workgroup of the persistent kernels is busy at once (the round-3
co-residency bug of the 13 .. 15-block border widths only showed there); those are checked on a 200-row sample, and every case is
run twice and must give the same bits.
Usage: python tests/scale/fuzz_parity.py [cases] [seed] [f1,f2,...]"""
import sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
from oracle import c_oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ONLY = None                                                     # optional: draw widths from this list ("a-b" = a range) only
if len(sys.argv) > 3:
    ONLY = sum([list(range(int(x.split("-")[0]), int(x.split("-")[1]) + 1)) if "-" in x else [int(x)] for x in sys.argv[3].split(",")], [])
FS = [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 50, 63, 64, 65, 80, 81, 96, 97, 112, 113, 127, 128, 129, 130, 143, 144, 145, 146,
      159, 160, 161, 176, 177, 200, 208, 209, 240, 241, 255, 256, 257, 258, 260]
worst = 0.0
t0 = time.perf_counter()
for case in range(cases):
    f = int(rng.choice(ONLY if ONLY else FS))
    bias = bool(rng.integers(2)) and f >= 2
    k = f - int(bias)
    m = int(rng.integers(max(2 * f, 40), 4 * f + 200))              # fixed side: enough rows for a decent Gramian
    n = int(rng.integers(20, 400)) if rng.integers(4) else int(rng.integers(1000, 3000))
    law = rng.choice(["poisson", "heavy", "mixed", "tiny"])
    rows, cols = [], []
    for u in range(n):
        if law == "poisson": d = rng.poisson(12)
        elif law == "heavy": d = int(rng.integers(30, min(m, 700)))
        elif law == "tiny": d = int(rng.integers(0, 4))
        else: d = int(rng.choice([0, 1, 15, 16, 17, 31, 32, 33, 34, 63, 64, 65, 100, min(m, 300)]))
        d = min(d, m)
        c = rng.choice(m, d, replace=False)
        rows.append(np.full(d, u)); cols.append(c)
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = (10 * np.log(1 + rng.integers(1, 8, len(rows)))).astype(np.float32)
    C = sp.csr_matrix((vals, (rows, cols)), shape=(n, m))
    model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias, seed=int(rng.integers(1 << 30)))
    Y = model.items
    if bias:                                                     # small biases: weights stay positive, the SPD kernels are used
        Y = Y.copy(); Y[:, 0] *= 0.5
    step = model.recompute_factors_bias if bias else model.recompute_factors
    print(f"case {case:3d}: f={f:3d} bias={int(bias)} n={n:3d} m={m:4d} law={law:7s} nnz={C.nnz:6d} ", end="", flush=True)
    got = step(Y, C, 0.1).astype(np.float64)
    assert np.array_equal(got, step(Y, C, 0.1)), "two runs on the same input differ"
    empty = np.diff(C.indptr) == 0
    assert not got[empty].any(), "rows without entries must be exactly zero"
    if n > 400:                                                  # the scalar C oracle on a sample of the rows
        pick = np.sort(rng.choice(n, 200, replace=False))
        C, got = C[pick], got[pick]
    want = c_oracle.half_step(Y, sp.csr_matrix((C.data.astype(np.float64), C.indices, C.indptr), shape=C.shape), 0.1, bias)
    err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30)
    tol = 5e-5 if f <= 144 else 1.5e-4
    flag = "" if err <= tol else "   <-- ABOVE TOLERANCE"
    worst = max(worst, err / tol)
    print(f"rel.err {err:.2e}{flag}")
print(f"{cases} cases in {time.perf_counter() - t0:.1f} s; worst error / tolerance = {worst:.2f}")
sys.exit(0 if worst <= 1.0 else 1)
