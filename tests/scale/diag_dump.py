"""Lab: run-to-run differences of the ACCUMULATED system (RS_DUMP variants of wmf_rowsplit.hip).  Usage: f"""
import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF
f = int(sys.argv[1])
rng = np.random.default_rng(5)
n, m, k = 1200, 700, f
deg = rng.integers(30, 700, n)
indptr = np.concatenate([[0], np.cumsum(deg)])
indices = np.concatenate([np.sort(rng.choice(m, d, replace=False)) for d in deg]).astype(np.int32)
data = (10 * np.log(1 + rng.integers(1, 8, indptr[-1]))).astype(np.float32)
C = sp.csr_matrix((data, indices, indptr), shape=(n, m))
model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=False)
Y = model.items.copy()
runs = [model.recompute_factors(Y, C, 0.1) for _ in range(3)]
for a in runs[1:]:
    dif = (a != runs[0])
    rows = np.flatnonzero(dif.any(axis=1))
    cols = np.flatnonzero(dif.any(axis=0))
    print("rows that differ", len(rows), "; columns that differ:", cols.tolist()[:60], "..." if len(cols) > 60 else "")
    if len(rows):
        r = rows[0]
        c = np.flatnonzero(dif[r])
        print("  row", r, "d", deg[r], "cols", c.tolist()[:20], "a", runs[0][r, c][:8], "b", a[r, c][:8])
import os
if len(sys.argv) > 2:
    os.makedirs("gpurun_out", exist_ok=True)
    np.save(sys.argv[2], np.stack(runs)[:, :, :208])
