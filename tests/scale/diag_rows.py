"""Ad-hoc: per-row errors of the test_half_step_vs_oracle_all_degree_classes case (k, bias), second half step."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import scipy.sparse as sp
from oracle import wmf_oracle as orc
from test_gpu_parity import ragged_matrix, as_f64
from recmodel_amd import WMF, _lib
k, bias = int(sys.argv[1]), bool(int(sys.argv[2]))
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = _lib.load(); lib.wmf_debug_set_flags(flags)
n, m_items = 2500, 600
C = ragged_matrix(n, m_items, seed=k + bias)
model = WMF(num_items=m_items, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
step_g = model.recompute_factors_bias if bias else model.recompute_factors
C64, CT = as_f64(C), C.T.tocsr()
Y = step_o(model.items, C64, 0.1)
want = step_o(Y, as_f64(CT), 0.1, out_dtype="float64")
for rep in range(3):
    got = step_g(Y, CT, 0.1)
    err = np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
    deg = np.diff(CT.indptr)
    bad = np.argsort(-err)[:8]
    b = Y[:, 0] if bias else np.zeros(n)
    print("rep", rep, "fro", np.linalg.norm(got - want) / np.linalg.norm(want))
    for r in bad:
        lo, hi = CT.indptr[r], CT.indptr[r + 1]
        w = CT.data[lo:hi] - (b[CT.indices[lo:hi]] if bias else 0)
        print(f"  row {r:4d} deg {deg[r]:4d} err {err[r]:.2e} min w_eff {w.min():.3f} nneg {(w < 0).sum()}")
