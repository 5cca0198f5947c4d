import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.')
from recmodel_amd import WMF, _lib
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
a = model.recompute_factors(Y, C, 0.1)
# the host entry point back-transforms what the kernel wrote (triangular); count the rows whose first 32 columns are not all zero
nz = np.abs(a[:, :32]).max(axis=1) > 0
print("rows with a mismatch counter != 0:", int(nz.sum()), "of", int((deg > 32).sum()), "heavy rows")
