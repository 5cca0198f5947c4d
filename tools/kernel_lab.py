"""Timing lab for the row-solve kernels on a real GPU (not a test): builds a cfg-shaped matrix,
prepares whitened factors once and times wmf_solve_rows per kernel slot, optionally under the
ablation flags of wmf_debug_set_flags.  Usage: python tools/kernel_lab.py cfg2 items 0,1,2,4,8"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
from recmodel_amd import _lib, synth
from recmodel_amd.engine import AlsEngine, _ptr, _stream
from recmodel_amd import WMF

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
side = sys.argv[2] if len(sys.argv) > 2 else "items"
flags = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _lib.load()
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda")
val = 10 * torch.log(1 + val)
eng = AlsEngine(n_users, n_items, k, bias, 0.1)
eng.set_interactions(ip, idx, val)
eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items)
eng.half_step("users"); eng.half_step("items"); eng.half_step("users")
torch.cuda.synchronize()
fixed = eng._other(side)
eng.prepare(fixed)
c = eng.csr[side]
print(f"{cfg} side={side} rows={c.n_rows} nnz={c.nnz} f={eng.f} bins rows={c.bin_rows.tolist()} nnz={c.bin_nnz.tolist()}")
for fl in flags:
    lib.wmf_debug_set_flags(fl)
    lib.wmf_profile_enable(0)
    for _ in range(2):
        _lib.check(lib.wmf_solve_rows(c._plan, _ptr(eng.V[fixed]), _ptr(eng.bias_vec[fixed]) if bias else None, _ptr(c.indptr),
                                      _ptr(c.indices), _ptr(c.values), c.n_rows, eng.f, eng.ld, _ptr(eng.g[side]), _ptr(eng.fail), _stream()))
    torch.cuda.synchronize()
    lib.wmf_profile_enable(1)
    for _ in range(reps):
        _lib.check(lib.wmf_solve_rows(c._plan, _ptr(eng.V[fixed]), _ptr(eng.bias_vec[fixed]) if bias else None, _ptr(c.indptr),
                                      _ptr(c.indices), _ptr(c.values), c.n_rows, eng.f, eng.ld, _ptr(eng.g[side]), _ptr(eng.fail), _stream()))
    torch.cuda.synchronize()
    lib.wmf_profile_enable(0)
    print(f"flags={fl}: " + ", ".join(f"{nm}={ms / reps:.3f}ms/{n // reps}" for nm, _, ms, n, _, _ in _lib.profile_table(lib)))
    lib.wmf_profile_reset()
    gnow = eng.g[side].clone()
    if fl == flags[0]:
        gref = gnow
    else:                                # kernel selections must agree (ablations will not)
        err = (gnow - gref).norm(dim=1) / gref.norm(dim=1).clamp_min(1e-30)
        print(f"    vs flags={flags[0]}: worst row {float(err.max()):.2e}, fro {float((gnow - gref).norm() / gref.norm()):.2e}, fails {int(eng.fail[0])}")
lib.wmf_debug_set_flags(0)
