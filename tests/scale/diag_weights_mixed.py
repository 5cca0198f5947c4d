"""Diagnostic (GPU, hand-run): rows that mix confidence weights from 1e-3 to 1e6 (the 'wide' matrix of
tests/test_gpu_parity.py::test_weight_range_of_the_class_surface) under different kernel selections.
python tests/scale/diag_weights_mixed.py [k] [bias]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
from recmodel_amd import WMF, _lib
from oracle import wmf_oracle as orc
import test_gpu_parity as T

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bias = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
lib = _lib.load()
rng = np.random.default_rng(1000 + k + bias)
n, m_items = 300, 6000
C, degs = T._weight_range_matrix(n, m_items, rng, "bias" if bias else "wide")
model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
Y = model.items.copy()
if bias:
    Y[:, 0] *= 0.5
step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
step_g = model.recompute_factors_bias if bias else model.recompute_factors
want = step_o(Y, T.as_f64(C), 0.1, out_dtype="float64")
ref32 = step_o(Y, C, 0.1).astype(np.float64)
den = np.linalg.norm(want, axis=1)
den[den == 0] = 1
er = np.linalg.norm(ref32 - want, axis=1) / den
# conditioning of the whitened system, per row (float64): 1 + lambda_max(V_u^T D V_u)
Yt = Y.astype(np.float64).copy()
bvec = np.zeros(m_items)
if bias:
    bvec, Yt[:, 0] = Yt[:, 0].copy(), 1.0
G = Yt.T @ Yt + 0.1 * np.eye(Y.shape[1])
Linv = np.linalg.inv(np.linalg.cholesky(G))
out = {}
for fl, name in ((0, "default"), (4096, "f32 register ring"), (33554432, "pivoted LU")):
    lib.wmf_debug_set_flags(fl)
    got = step_g(Y, C, 0.1).astype(np.float64)
    lib.wmf_debug_set_flags(0)
    out[name] = np.linalg.norm(got - want, axis=1) / den
print(f"k={k} bias={bias}")
print("   d      wmax     cond_w    " + "  ".join(f"{nm:>18s}" for nm in out) + "        numpy-f32")
for r in np.argsort(degs, kind="stable"):
    if degs[r] < 30 or degs[r] > 140:
        continue
    lo, hi = C.indptr[r], C.indptr[r + 1]
    w = C.data[lo:hi].astype(np.float64) - bvec[C.indices[lo:hi]]
    Vu = Yt[C.indices[lo:hi]] @ Linv.T
    cond = 1 + np.linalg.eigvalsh(Vu.T @ (Vu * w[:, None]))[-1]
    print(f"{degs[r]:5d}  {C.data[lo:hi].max():9.1e}  {cond:9.1e}    " + "  ".join(f"{out[nm][r]:18.2e}" for nm in out) + f"   {er[r]:14.2e}")
