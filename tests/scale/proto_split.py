"""Numerical prototype (CPU, NumPy): what would split low-precision MFMA products do to a half step?
The accumulations S = V_u V_u^T (rows with few entries) and B = I + V_u^T D V_u (heavy rows) are computed as
  f32     plain float32 products                               (what the kernels do today)
  bf16x6  three bf16 parts per operand, six products            (a1b1 a1b2 a2b1 a1b3 a2b2 a3b1)
  bf16x3  two bf16 parts, three products
  fp16x3  two fp16 parts after a power-of-two scaling, three products
and the half step is compared with the float64 oracle.  Not shipped."""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/scale')
from oracle import wmf_oracle as orc
from proto_whiten import synth, whiten, relerr
f32 = np.float32

def bf16(x):
    u = np.asarray(x, dtype=f32).view(np.uint32).astype(np.uint64)
    u = ((u + ((u >> 16) & 1) + 0x7fff) >> 16) << 16
    return u.astype(np.uint32).view(f32)

def parts(x, mode):
    x = np.asarray(x, f32)
    if mode.startswith("bf16"):
        a1 = bf16(x); r = (x - a1).astype(f32); a2 = bf16(r); a3 = bf16((r - a2).astype(f32))
        return [a1, a2, a3][: 3 if mode == "bf16x6" else 2]
    a1 = x.astype(np.float16).astype(f32); a2 = (x - a1).astype(f32).astype(np.float16).astype(f32)
    return [a1, a2]

def prod(A, B, mode):
    """A @ B.T with the products of `mode`, float32 accumulation, small terms first."""
    if mode == "f32":
        return (A @ B.T).astype(f32)
    scale = f32(1.0)
    if mode == "fp16x3":
        amax = max(np.abs(A).max(), np.abs(B).max(), 1e-30)
        scale = f32(2.0 ** np.floor(np.log2(4096.0 / amax)))            # largest element near 2^11: hi parts far from overflow
    pa, pb = parts(A * scale, mode), parts(B * scale, mode)
    n = len(pa)
    terms = sorted([(i, j) for i in range(n) for j in range(n) if i + j <= n - 1], key=lambda t: -(t[0] + t[1]))
    acc = np.zeros((A.shape[0], B.shape[0]), f32)
    for i, j in terms:
        acc = (acc + (pa[i] @ pb[j].T).astype(f32)).astype(f32)
    return (acc / (scale * scale)).astype(f32)

def half_step(Y, C, lam, bias, mode, thresh=33):
    V, Linv, b, cond = whiten(Y, lam, bias)
    n, f = C.shape[0], Y.shape[1]
    G = np.zeros((n, f), f32)
    for r in range(n):
        lo, hi = C.indptr[r], C.indptr[r + 1]
        if hi == lo: continue
        idx = C.indices[lo:hi]; w = C.data[lo:hi].astype(f32)
        if bias: w = (w - b[idx]).astype(f32)
        Vu = V[idx]; d = len(idx); p = (w + 1).astype(f32)
        if d < thresh:
            S = prod(Vu, Vu, mode)
            M = (np.eye(d, dtype=f32) + w[:, None] * S).astype(f32)
            G[r] = (Vu.T @ np.linalg.solve(M, p).astype(f32)).astype(f32)
        else:
            if (w < 0).any():
                B = (np.eye(f, dtype=f32) + Vu.T @ (Vu * w[:, None])).astype(f32)
            else:
                X = (Vu * np.sqrt(w)[:, None]).astype(f32)
                B = (np.eye(f, dtype=f32) + prod(X.T.copy(), X.T.copy(), mode)).astype(f32)
            G[r] = np.linalg.solve(B, (Vu.T @ p).astype(f32)).astype(f32)
    return (G @ Linv).astype(f32)

def fro(a, b):
    return np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)

if __name__ == "__main__":
    for k, bias in ((64, False), (128, False), (128, True)):
        n, m = 1500, 400
        C = synth(n, m, 14, 3); CT = C.T.tocsr()
        Y = orc.init_items(m, k, bias)
        step = orc.recompute_factors_bias if bias else orc.recompute_factors
        for it in range(2):
            for side, (F, M) in (("users", (Y, C)), ("items", (None, CT))):
                if side == "items": F = Xf
                ref = step(F, M.astype(np.float64), 0.1, out_dtype='float64')
                line = f"k={k} bias={int(bias)} it={it} {side:5s}"
                for mode in ("f32", "bf16x6", "fp16x3", "bf16x3"):
                    X = half_step(F, M, 0.1, bias, mode)
                    line += f" | {mode} fro {fro(X, ref):.1e} row {relerr(X, ref).max():.1e}"
                print(line, flush=True)
                if side == "users": Xf = ref.astype(f32)
                else: Y = ref.astype(f32)
