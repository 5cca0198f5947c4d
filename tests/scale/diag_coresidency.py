"""Diagnostic (GPU, hand-run): does a wide-row half step stay bit-identical while OTHER kernels share the compute units?
A background thread keeps a side stream busy (elementwise kernels, then f32 GEMMs); the half step runs five times meanwhile.
Usage: python tests/scale/diag_coresidency.py f bias"""
import sys, threading
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from recmodel_amd import WMF

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
quiet = step(Y, C, 0.1)
assert np.array_equal(quiet, step(Y, C, 0.1))
for kind in ("elementwise", "gemm"):
    stop = threading.Event()
    launched = [0]
    def noise():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x = torch.ones(1 << 26, device="cuda")
            a = torch.randn(2048, 2048, device="cuda"); b = torch.randn(2048, 2048, device="cuda")
            while not stop.is_set():
                for _ in range(20):
                    if kind == "elementwise": x.mul_(1.0000001)
                    else: torch.mm(a, b)
                    launched[0] += 1
                s.synchronize()
    t = threading.Thread(target=noise); t.start()
    bad = 0
    for _ in range(5):
        got = step(Y, C, 0.1)
        bad += int((np.abs(got - quiet).max(axis=1) > 0).sum())
    stop.set(); t.join()
    print(f"f={f} bias={int(bias)} noise={kind}: {launched[0]} noise kernels meanwhile; rows that differ from the quiet run: {bad}")
