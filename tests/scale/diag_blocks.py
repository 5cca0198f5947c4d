"""Diagnostic (GPU, hand-run): per 16-column block error of the wrong rows of a heavy-row case.  Usage: f bias"""
import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
from oracle import c_oracle
f, bias = int(sys.argv[1]), bool(int(sys.argv[2]))
rng = np.random.default_rng(5)
n, m, k = 1200, 700, f - int(bias)
deg = rng.integers(30, 700, n)
indptr = np.concatenate([[0], np.cumsum(deg)])
indices = np.concatenate([np.sort(rng.choice(m, d, replace=False)) for d in deg]).astype(np.int32)
data = (10 * np.log(1 + rng.integers(1, 8, indptr[-1]))).astype(np.float32)
C = sp.csr_matrix((data, indices, indptr), shape=(n, m))
model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
Y = model.items.copy()
if bias: Y[:, 0] *= 0.5
step = model.recompute_factors_bias if bias else model.recompute_factors
want = c_oracle.half_step(Y, sp.csr_matrix((C.data.astype(np.float64), C.indices, C.indptr), shape=C.shape), 0.1, bias)
got = step(Y, C, 0.1).astype(np.float64)
# the host entry point returns the reference's column order (bias first); the kernel's order has the bias LAST
if bias: got, want = np.roll(got, -1, axis=1), np.roll(want, -1, axis=1)
e = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
bad = np.flatnonzero(e > 1e-3)
print("wrong rows", len(bad), "first", bad[:20].tolist())
nb = (f + 15) // 16
np.set_printoptions(linewidth=250, precision=1)
for r in bad[:12]:
    be = [np.abs(got[r, 16 * b: 16 * b + 16] - want[r, 16 * b: 16 * b + 16]).max() for b in range(nb)]
    print(f"row {r:4d} d={deg[r]:3d} rel {e[r]:.1e} |want|max {np.abs(want[r]).max():.2f} per-block max abs err:", " ".join(f"{x:.0e}" for x in be))
good = np.flatnonzero(e <= 1e-3)
print("position of wrong rows modulo 512:", np.bincount(bad % 512 // 64, minlength=8).tolist(), " (all rows:", np.bincount(np.arange(n) % 512 // 64, minlength=8).tolist(), ")")
print("degree of wrong rows: mean", deg[bad].mean(), "good:", deg[good].mean())
