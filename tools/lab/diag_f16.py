"""GPU diagnosis (not a test): per-row error of the heavy-row kernel variants against the float64 oracle on the ragged
matrix of tests/test_gpu_parity.py.  Usage: python tools/lab/diag_f16.py [k] [bias]"""
import sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import wmf_oracle as orc
from recmodel_amd import WMF, _lib
src = open('tests/test_gpu_parity.py').read()
ns = {'np': np, 'sp': sp}
exec(src[src.index('def as_f64'):src.index('@pytest.mark.parametrize("k,bias", [(16, False)')], ns)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 128
bias = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
n, m = 2500, 600
C = ns['ragged_matrix'](n, m, seed=k + bias)
model = WMF(num_items=m, num_users=n, dim=k, gamma=0.1, weighted=True, bias=bias)
step_o = orc.recompute_factors_bias if bias else orc.recompute_factors
step_g = model.recompute_factors_bias if bias else model.recompute_factors
lib = _lib.load()
C64, CT = ns['as_f64'](C), C.T.tocsr()
for name, (Y, mat) in (("users", (model.items, C)), ("items", (step_o(model.items, C64, 0.1), CT))):
    want = step_o(Y, ns['as_f64'](mat), 0.1, out_dtype="float64")
    deg = np.diff(mat.indptr)
    for flags in (0, 8192, 4096):
        lib.wmf_debug_set_flags(flags)
        got = step_g(Y, mat, 0.1).astype(np.float64)
        lib.wmf_debug_set_flags(0)
        rel = np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-30)
        heavy = deg > 32
        fro = np.linalg.norm(got - want) / np.linalg.norm(want)
        w = np.argsort(-rel * heavy)[:4]
        print(f"{name} flags={flags}: fro {fro:.2e} heavy rows {heavy.sum()} worst heavy {[(int(i), int(deg[i]), float(f'{rel[i]:.1e}')) for i in w]} "
              f"light max {rel[~heavy].max():.1e} heavy fro {np.linalg.norm((got - want)[heavy]) / np.linalg.norm(want[heavy]):.2e}")
