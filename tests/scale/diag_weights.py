"""Diagnostic (GPU, hand-run): per-row error of the HIP half step against the float64 oracle under scaled confidence
weights.  python tests/scale/diag_weights.py [k] [bias]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import scipy.sparse as sp
from recmodel_amd import WMF
from oracle import wmf_oracle as orc

k = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bias = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
rng = np.random.default_rng(5)
n, m_items = 120, 6000
degs = [1, 2, 4, 8, 9, 12, 16, 17, 24, 32, 33, 64, 100, 400, 5000] * 8
degs = degs[:n]
indptr = np.concatenate([[0], np.cumsum(degs)])
indices = np.concatenate([np.sort(rng.choice(m_items, d, replace=False)) for d in degs]).astype(np.int32)
model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
Y = model.items.copy()
if bias:
    Y[:, 0] *= 0.5
step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
step_g = model.recompute_factors_bias if bias else model.recompute_factors
for scale in (1.0, 10.0, 100.0, 1e3, 1e4, 1e5):
    # block b of 15 rows: weights = scale * (7 .. 19); all rows the same structure
    w = (scale * 10 * np.log(1 + rng.integers(1, 6, indptr[-1]))).astype(np.float32)
    C = sp.csr_matrix((w, indices, indptr), shape=(n, m_items))
    C64 = sp.csr_matrix((w.astype(np.float64), indices, indptr), shape=(n, m_items))
    want = step_o(Y, C64, 0.1, out_dtype="float64")
    ref32 = step_o(Y, C, 0.1).astype(np.float64)
    got = step_g(Y, C, 0.1).astype(np.float64)
    den = np.linalg.norm(want, axis=1)
    eg = np.linalg.norm(got - want, axis=1) / den
    er = np.linalg.norm(ref32 - want, axis=1) / den
    print(f"k={k} bias={bias} scale={scale:g}")
    for j in range(15):
        rows = np.arange(j, n, 15)
        print(f"   d={degs[j]:5d}  gpu max {eg[rows].max():.2e} med {np.median(eg[rows]):.2e}   numpy-f32 max {er[rows].max():.2e} med {np.median(er[rows]):.2e}")
