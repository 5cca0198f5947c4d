"""Size-independent parity check at full benchmark scale (not a pytest: cfg3 needs tens of GB of host memory for the
checks): after one ALS iteration on the synthetic matrix of a bench configuration, sampled rows of both sides must satisfy
their own normal equations  (G + U^T diag(w) U) x = U^T (w + 1)  computed in float64 on the host from the factors the
device produced (RecModel/wmf_model.py:233-239), and rows without entries must be zero.
Usage: python tools/verify_at_scale.py cfg3 [samples]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from recmodel_amd import WMF, synth
from recmodel_amd.engine import AlsEngine

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda")
w = 10 * torch.log(1 + val)
eng = AlsEngine(n_users, n_items, k, bias, 0.1)
eng.set_interactions(ip, idx, w)
eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items)
import scipy.sparse as sp


def check(side, X, Y, indptr, indices, weights, n_rows):
    """rows of X (just updated, float64) against the fixed side Y through their CSR"""
    f = Y.shape[1]
    Yt, bvec = Y.copy(), np.zeros(Y.shape[0])
    if bias:
        bvec = Y[:, 0].copy(); Yt[:, 0] = 1.0
    G = Yt.T @ Yt + 0.1 * np.eye(f)
    rng = np.random.default_rng(0)
    deg = np.diff(indptr)
    pick = np.concatenate([rng.choice(n_rows, ns, replace=False), np.argsort(deg)[-20:], np.flatnonzero(deg <= 1)[:20]])
    by_class = {}
    for u in pick:
        lo, hi = indptr[u], indptr[u + 1]
        U, wu = Yt[indices[lo:hi]], weights[lo:hi] - bvec[indices[lo:hi]]
        A, b = G + U.T @ (U * wu[:, None]), (wu + 1) @ U
        res = np.linalg.norm(A @ X[u] - b) / max(np.linalg.norm(b), 1e-30) if hi > lo else np.abs(X[u]).max()
        c = "d=0" if hi == lo else ("d<=16" if hi - lo <= 16 else ("d<=32" if hi - lo <= 32 else "d>32"))
        by_class[c] = max(by_class.get(c, 0.0), res)
    print(f"{side} rows, worst relative residual of the normal equations by degree class:", {c: f"{v:.2e}" for c, v in by_class.items()})
    assert max(by_class.values()) <= 5e-4, by_class


iph, idxh, wh = ip.cpu().numpy(), idx.cpu().numpy(), w.cpu().numpy().astype(np.float64)
C = sp.csr_matrix((wh, idxh, iph), shape=(n_users, n_items))
CT = C.T.tocsr()
t0 = time.perf_counter()
eng.half_step("users")
X1 = eng.get_factors("users").astype(np.float64)
eng.half_step("items")
Y2 = eng.get_factors("items").astype(np.float64)
eng.half_step("users")
X3 = eng.get_factors("users").astype(np.float64)
torch.cuda.synchronize()
print(f"{cfg}: three half steps (incl. copying the factors to the host) in {time.perf_counter() - t0:.2f} s")
eng.check_numerics()
check("item", Y2, X1, CT.indptr, CT.indices, CT.data, n_items)
check("user", X3, Y2, C.indptr, C.indices, C.data, n_users)
print("OK")
