"""Numerical prototype (CPU, numpy): fp32 whitened-Woodbury / whitened-direct vs the fp64 oracle.
Not shipped; used to choose the formulation and the parity tolerance."""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from oracle import wmf_oracle as orc
f32 = np.float32

def synth(n, m, dbar, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = np.maximum(rng.poisson(dbar, n), 1)
    pop = 1.0 / np.arange(1, m + 1) ** 0.8; pop /= pop.sum(); cdf = np.cumsum(pop)
    rows = np.repeat(np.arange(n), deg)
    cols = np.searchsorted(cdf, rng.random(len(rows))).clip(0, m - 1)
    key = np.unique(rows.astype(np.int64) * m + cols)
    rows, cols = key // m, key % m
    vals = (1 + rng.geometric(0.5, len(rows))).astype(f32)
    C = sp.csr_matrix((vals, (rows, cols)), shape=(n, m))
    C.data = (10 * np.log(1 + C.data)).astype(f32)
    return C

def whiten(Y, lam, bias):
    Yt = Y.copy()
    b = None
    if bias:
        b = Yt[:, 0].copy(); Yt[:, 0] = 1
    G = Yt.astype(np.float64).T @ Yt.astype(np.float64) + lam * np.eye(Y.shape[1])
    L = np.linalg.cholesky(G)
    Linv = np.linalg.inv(L)
    V = (Yt @ Linv.T.astype(f32)).astype(f32)           # fp32 GEMM
    return V, Linv.astype(f32), b, np.linalg.cond(G)

def half_step_whitened(Y, C, lam, bias, thresh=None):
    V, Linv, b, cond = whiten(Y, lam, bias)
    n, f = C.shape[0], Y.shape[1]
    Gout = np.zeros((n, f), f32)
    for r in range(n):
        lo, hi = C.indptr[r], C.indptr[r + 1]
        if hi == lo: continue
        idx = C.indices[lo:hi]; w = C.data[lo:hi].astype(f32)
        if bias: w = (w - b[idx]).astype(f32)
        Vu = V[idx]; d = len(idx); p = (w + 1).astype(f32)
        if d < (thresh or f):
            S = (Vu @ Vu.T).astype(f32)
            M = (np.eye(d, dtype=f32) + w[:, None] * S).astype(f32)
            c = np.linalg.solve(M, p).astype(f32)
            g = (Vu.T @ c).astype(f32)
        else:
            B = (np.eye(f, dtype=f32) + Vu.T @ (Vu * w[:, None])).astype(f32)
            g = np.linalg.solve(B, (Vu.T @ p).astype(f32)).astype(f32)
        Gout[r] = g
    X = (Gout @ Linv).astype(f32)
    return X, cond

def relerr(a, b):
    num = np.linalg.norm(a.astype(np.float64) - b, axis=1); den = np.linalg.norm(b, axis=1) + 1e-30
    return (num / den)

for k, bias in ((64, False), (128, False), (128, True)):
    n, m = 3000, 1500
    C = synth(n, m, 12, 3); CT = C.T.tocsr()
    C64 = C.astype(np.float64); CT64 = CT.astype(np.float64)
    Y = orc.init_items(m, k, bias)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    for it in range(3):
        Xref = step(Y, C64, 0.1, out_dtype='float64')
        X, cond = half_step_whitened(Y, C, 0.1, bias)
        e = relerr(X, Xref); print(f"k={k} bias={bias} it={it} users: cond(G)={cond:.3g} rel-err max={e.max():.2e} med={np.median(e):.2e}  absmax={np.abs(X-Xref).max():.2e} xmax={np.abs(Xref).max():.2f}")
        Xf = Xref.astype(f32)
        Yref = step(Xf, CT64, 0.1, out_dtype='float64')
        Y2, cond = half_step_whitened(Xf, CT, 0.1, bias)
        e = relerr(Y2, Yref); print(f"k={k} bias={bias} it={it} items: cond(G)={cond:.3g} rel-err max={e.max():.2e} med={np.median(e):.2e}  absmax={np.abs(Y2-Yref).max():.2e}")
        Y = Yref.astype(f32)

print("--- reference's own fp32-count path (fp32 compute via numpy/LAPACK sgesv) vs its fp64 path")
for k, bias in ((128, True), (128, False)):
    n, m = 3000, 1500
    C = synth(n, m, 12, 3); C64 = C.astype(np.float64)
    Y = orc.init_items(m, k, bias)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    Xref = step(Y, C64, 0.1, out_dtype='float64')
    X32 = step(Y, C, 0.1)
    e = relerr(X32, Xref); print(f"k={k} bias={bias} it=0 users: ref-fp32 rel-err max={e.max():.2e} med={np.median(e):.2e}")
