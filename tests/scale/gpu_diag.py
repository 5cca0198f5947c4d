"""Kernel-by-kernel diagnostic on a real GPU: prints error norms of each C-ABI building block
against NumPy (fp64) so a failing parity test can be localised from one gpurun call.
Usage: python tests/scale/gpu_diag.py [k] [bias]"""
import ctypes, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from recmodel_amd import _lib, synth
from recmodel_amd.engine import _ptr, _stream
from oracle import wmf_oracle as orc

lib = _lib.load()
dev = "cuda"
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bias = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, m, dbar = (int(sys.argv[3]) if len(sys.argv) > 3 else 3000), (int(sys.argv[4]) if len(sys.argv) > 4 else 700), 12
f = k + bias
ld = lib.wmf_ld_for(f)
print(f"== diag k={k} bias={bias} f={f} ld={ld} n={n} m={m}")
rng = np.random.default_rng(0)
Y = rng.random((m, f)).astype(np.float32)
Yd = torch.zeros(m, ld, device=dev); Yd[:, :f] = torch.from_numpy(Y).to(dev)
ws = torch.empty(int(lib.wmf_gram_workspace_bytes(f)), dtype=torch.uint8, device=dev)
G = torch.zeros(f * f, dtype=torch.float64, device=dev)
_lib.check(lib.wmf_gram(_ptr(Yd), m, f, ld, bias, _ptr(G), _ptr(ws), _stream()))
torch.cuda.synchronize()
Yt = Y.astype(np.float64).copy()
if bias: Yt[:, 0] = 1
Gref = Yt.T @ Yt
Gh = G.cpu().numpy().reshape(f, f)
print("gram      rel err", np.abs(Gh - Gref).max() / np.abs(Gref).max(), "sym", np.abs(Gh - Gh.T).max())

Ww = torch.zeros(f, ld, device=dev); Wu = torch.zeros(f, ld, device=dev); info = torch.zeros(4, dtype=torch.int32, device=dev)
lam = 0.1
_lib.check(lib.wmf_factorize(_ptr(G), f, ld, lam, _ptr(Ww), _ptr(Wu), _ptr(info), _ptr(ws), _stream()))
torch.cuda.synchronize()
L = np.linalg.cholesky(Gh + lam * np.eye(f)); Linv = np.linalg.inv(L)
Wwh, Wuh = Ww.cpu().numpy(), Wu.cpu().numpy()
print("factorize info", info[0].item(), "Wu err", np.abs(Wuh[:, :f] - Linv).max() / np.abs(Linv).max(),
      "Ww err", np.abs(Wwh[:, :f] - Linv.T).max() / np.abs(Linv).max(), "pad", np.abs(Wwh[:, f:]).max() if ld > f else 0.0)

V = torch.full((m, ld), 7.0, device=dev); bvec = torch.zeros(m, device=dev)
_lib.check(lib.wmf_row_transform(_ptr(Yd), m, f, ld, _ptr(Ww), bias, _ptr(V), _ptr(bvec) if bias else None, _stream()))
torch.cuda.synchronize()
Vref = Yt @ Linv.T
Vh = V.cpu().numpy()
print("transform rel err", np.abs(Vh[:, :f] - Vref).max() / np.abs(Vref).max(), "pad", np.abs(Vh[:, f:]).max() if ld > f else 0.0,
      "bias err", (np.abs(bvec.cpu().numpy() - Y[:, 0]).max() if bias else 0.0))

# CSR with a spread of degrees
ip, idx, val = synth.make_counts(n, m, dbar, 3)
C = synth.to_scipy(ip, idx, val, (n, m)).tolil()
special = {0: 0, 1: 1, 2: 16, 3: 17, 4: 32, 5: 33, 6: 100, 7: min(m, 400), 8: 2}
for r, d in special.items():
    cols = np.sort(rng.choice(m, d, replace=False))
    C.rows[r] = list(cols); C.data[r] = list((2 + rng.integers(0, 4, d)).astype(np.float32))
C = C.tocsr(); C.data = (10 * np.log(1 + C.data)).astype(np.float32)
C.data[C.indptr[8]] = 0.0   # stored zero
ipd = torch.from_numpy(C.indptr.astype(np.int64)).to(dev); idxd = torch.from_numpy(C.indices.astype(np.int32)).to(dev)
vald = torch.from_numpy(C.data).to(dev)
hp = np.ascontiguousarray(C.indptr, dtype=np.int64)
plan = ctypes.c_void_p(); _lib.check(lib.wmf_plan_create(hp.ctypes.data_as(ctypes.c_void_p), n, f, ctypes.byref(plan)))
st4 = np.zeros(8, np.int64); lib.wmf_plan_stats(plan, st4.ctypes.data_as(ctypes.c_void_p)); print("plan bins", st4)
g = torch.full((n, ld), 5.0, device=dev); fail = torch.zeros(4, dtype=torch.int32, device=dev)
t0 = time.time()
_lib.check(lib.wmf_solve_rows(plan, _ptr(V), _ptr(bvec) if bias else None, _ptr(ipd), _ptr(idxd), _ptr(vald), n, f, ld, _ptr(g), _ptr(fail), _stream()))
torch.cuda.synchronize(); print("solve time", time.time() - t0, "fail", fail[0].item())
X = torch.zeros(n, ld, device=dev)
_lib.check(lib.wmf_row_transform(_ptr(g), n, f, ld, _ptr(Wu), 0, _ptr(X), None, _stream()))
torch.cuda.synchronize()
Xh = X.cpu().numpy()[:, :f]
step = orc.recompute_factors_bias if bias else orc.recompute_factors
import scipy.sparse as sp
Xref = step(Y, sp.csr_matrix((C.data.astype(np.float64), C.indices, C.indptr), shape=C.shape), lam, out_dtype='float64')
deg = np.diff(C.indptr)
num = np.linalg.norm(Xh - Xref, axis=1); den = np.linalg.norm(Xref, axis=1) + 1e-30
rel = num / den
for name, mask in (("d==0", deg == 0), ("1..16", (deg > 0) & (deg <= 16)), ("17..32", (deg > 16) & (deg <= 32)), (">32", deg > 32)):
    if mask.any():
        worst = np.where(mask)[0][np.argmax(np.where(deg[mask] == 0, num[mask], rel[mask]))]
        print(f"rows {name:7s} count {mask.sum():6d} max rel err {rel[mask].max() if name!='d==0' else num[mask].max():.3e} (row {worst}, deg {deg[worst]})")
print("special rows:", {r: (int(deg[r]), float(f"{rel[r]:.2e}")) for r in special})
print("overall rel Frobenius", np.linalg.norm(Xh - Xref) / np.linalg.norm(Xref), "nan:", np.isnan(Xh).sum())

# host-level entry point
Xhost = np.empty((n, f), np.float32)
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
rc = lib.wmf_recompute_factors_host(vp(Y), m, f, bias, vp(hp), vp(np.ascontiguousarray(C.indices, np.int32)), vp(C.data), n, lam, vp(Xhost))
print("host entry rc", rc, lib.wmf_last_error() if rc else "", "rel Frobenius", np.linalg.norm(Xhost - Xref) / np.linalg.norm(Xref))

# eval / predict
items = torch.zeros(m, ld, device=dev); items[:, :f] = torch.from_numpy(Y).to(dev)
users = X
ews = torch.empty(int(lib.wmf_eval_workspace_bytes()), dtype=torch.uint8, device=dev); out3 = torch.zeros(3, dtype=torch.float64, device=dev)
_lib.check(lib.wmf_eval_sqerr(_ptr(users), _ptr(items), f, ld, bias, _ptr(ipd), _ptr(idxd), _ptr(vald), n, _ptr(out3), _ptr(ews), _stream()))
torch.cuda.synchronize()
mse_ref = orc.eval_prec(Xh, Y, C, bool(bias)); mae_ref = orc.eval_prec(Xh, Y, C, bool(bias), "mae")
o = out3.cpu().numpy(); print("eval mse", o[0] / o[2], "ref", mse_ref, "mae", o[1] / o[2], "ref", mae_ref, "count", o[2], "ref", (C.data != 0).sum())
lib.wmf_plan_destroy(plan)
