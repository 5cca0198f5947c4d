"""Timing lab for the dense shared-work kernels on a real GPU (not a test): Gramian and row transform of an [m, f] factor
block.  Usage: python tools/dense_lab.py m k bias reps [flags,...]"""
import sys
import torch
sys.path.insert(0, '.')
from recmodel_amd import _lib
from recmodel_amd.engine import HipKernels

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 128
bias = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
flags = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "0").split(",")]
lib = _lib.load()
K = HipKernels()
f = k + (2 if bias else 0) - (1 if bias else 0)            # the engine's f: k + 1 with biases
ld = K.ld_for(f)
dev = torch.device("cuda")
torch.manual_seed(5)
X = torch.rand(m, ld, device=dev) * 0.1
X[:, f:] = 0
G = torch.zeros(f, f, dtype=torch.float64, device=dev)
ws = torch.empty(K.gram_workspace_bytes(f), dtype=torch.uint8, device=dev)
W = torch.triu(torch.randn(f, ld, device=dev) * 0.05)          # (the whitening matrices are triangular)
ldv = K.whitened_row_floats(f, ld, bias)
V = torch.empty(m, ldv, device=dev)
pairs = torch.empty(m, 2, device=dev) if (bias and ldv != ld) else torch.empty(m, device=dev)
out2 = torch.empty(m, ld, device=dev)
Gref = None
for fl in flags:
    lib.wmf_debug_set_flags(fl)
    lib.wmf_profile_enable(0)
    K.gram(X, m, f, ld, bias, G, ws)
    K.row_transform(X, m, f, ld, W, bias, V, pairs if bias else None)
    K.row_transform(X, m, f, ld, W, False, out2, None)
    torch.cuda.synchronize()
    lib.wmf_profile_enable(1)
    for _ in range(reps):
        K.gram(X, m, f, ld, bias, G, ws)
        K.row_transform(X, m, f, ld, W, bias, V, pairs if bias else None)
        K.row_transform(X, m, f, ld, W, False, out2, None)
    torch.cuda.synchronize()
    lib.wmf_profile_enable(0)
    print(f"flags={fl}: " + ", ".join(f"{nm}={ms / max(n, 1):.3f}ms x{n // reps}" for nm, _, ms, n, _, _ in _lib.profile_table(lib)))
    lib.wmf_profile_reset()
    if Gref is None:
        Gref = G.clone()
        G64 = (X[:1_000_000, :f].double().T @ X[:1_000_000, :f].double()) if not bias else None
    else:
        print(f"    gram vs flags={flags[0]}: {float((G - Gref).abs().max() / Gref.abs().max()):.2e}")
lib.wmf_debug_set_flags(0)
