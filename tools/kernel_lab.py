"""Timing lab for the row-solve kernels on a real GPU (not a test): builds a cfg-shaped matrix,
prepares whitened factors once and times wmf_solve_rows per kernel slot, optionally under the
ablation flags of wmf_debug_set_flags.  Usage: python tools/kernel_lab.py cfg2 items 0,1,2,4,8"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
from recmodel_amd import _lib, synth
from recmodel_amd.engine import AlsEngine, _ptr, _stream
from recmodel_amd import WMF

import os
ZIPF = float(os.environ.get("LAB_ZIPF", "0"))          # item popularity exponent (0 = uniform)
HIST = os.environ.get("LAB_HIST") == "1"               # print the distribution of tr E = sum_e w_e |v_e|^2 over the side's rows
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
side = sys.argv[2] if len(sys.argv) > 2 else "items"
flags = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _lib.load()
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda", zipf_a=ZIPF)
val = 10 * torch.log(1 + val)
eng = AlsEngine(n_users, n_items, k, bias, 0.1)
eng.set_interactions(ip, idx, val)
eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items)
eng.half_step("users"); eng.half_step("items"); eng.half_step("users")
torch.cuda.synchronize()
fixed = eng._other(side)
eng.prepare(fixed)
c = eng.csr[side]
print(f"{cfg} zipf={ZIPF} side={side} rows={c.n_rows} nnz={c.nnz} f={eng.f} bins rows={c.bin_rows.tolist()} nnz={c.bin_nnz.tolist()} "
      f"iteration candidates: rows={c.rows_iter} nnz={c.nnz_iter}")
if HIST:
    # tr E of every row: the bound the iteration kernel (csrc/wmf_iter.hip) decides on.  Whitened rows: the packed body, plus
    # the border feature of the split layout (first float of the pairs); weights: minus the fixed side's bias there.
    V = eng.V[fixed]
    n2 = (V.double() ** 2).sum(1)
    w = c.values.double()
    if bias and eng.split:
        n2 = n2 + eng.bias_vec[fixed][:, 0].double() ** 2
        w = w - eng.bias_vec[fixed][:, 1].double()[c.indices.long()]
    elif bias:
        w = w - eng.bias_vec[fixed].double()[c.indices.long()]
    deg = c.indptr[1:] - c.indptr[:-1]
    rows_of = torch.repeat_interleave(torch.arange(c.n_rows, device=w.device), deg)
    contrib = w * n2[c.indices.long()]
    tau_p = torch.zeros(c.n_rows, dtype=torch.float64, device=w.device).index_add_(0, rows_of, contrib.clamp_min(0))
    tau_n = torch.zeros(c.n_rows, dtype=torch.float64, device=w.device).index_add_(0, rows_of, (-contrib).clamp_min(0))
    for name, sel in (("rows with 33+ entries", deg > 32), ("rows with <= 32 entries", (deg > 0) & (deg <= 32))):
        if int(sel.sum()) == 0:
            continue
        t = tau_p[sel]
        qs = torch.quantile(t[torch.randperm(t.numel(), device=t.device)[:2_000_000]], torch.tensor([0.01, 0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64, device=t.device))
        print(f"  tr E, {name} ({int(sel.sum())}): q01 {qs[0]:.3g}  median {qs[1]:.3g}  q90 {qs[2]:.3g}  q99 {qs[3]:.3g}  q99.9 {qs[4]:.3g}  max(sample) {qs[5]:.3g};"
              f"  share <= 0.06: {float((t <= 0.06).double().mean()):.4f}, <= 0.5: {float((t <= 0.5).double().mean()):.4f}, <= 3: {float((t <= 3).double().mean()):.4f};"
              f"  negative part max {float(tau_n[sel].max()):.3g}")
for fl in flags:
    lib.wmf_debug_set_flags(fl)
    lib.wmf_profile_enable(0)
    for _ in range(2):
        _lib.check(lib.wmf_solve_rows_ex(c._plan, _ptr(eng.V[fixed]), _ptr(eng.bias_vec[fixed]) if bias else None, _ptr(c.indptr),
                                      _ptr(c.indices), _ptr(c.values), c.n_rows, eng.f, eng.ld, _ptr(eng.g[side]), _ptr(eng.fail), eng.solve_flags, _stream()))
    torch.cuda.synchronize()
    lib.wmf_profile_enable(1)
    for _ in range(reps):
        _lib.check(lib.wmf_solve_rows_ex(c._plan, _ptr(eng.V[fixed]), _ptr(eng.bias_vec[fixed]) if bias else None, _ptr(c.indptr),
                                      _ptr(c.indices), _ptr(c.values), c.n_rows, eng.f, eng.ld, _ptr(eng.g[side]), _ptr(eng.fail), eng.solve_flags, _stream()))
    torch.cuda.synchronize()
    lib.wmf_profile_enable(0)
    print(f"flags={fl}: " + ", ".join(f"{nm}={ms / reps:.3f}ms/{n // reps}" for nm, _, ms, n, _, _ in _lib.profile_table(lib)))
    st = c.iter_stats() // max(1, reps + 2)
    if st.sum():
        print(f"    iteration kernel per launch: solved {st[0]}, handed back {st[1]}, applications of E per solved row "
              f"{st[2] / max(1, st[0]):.2f}, on the Chebyshev recurrence {st[3]}")
    lib.wmf_profile_reset()
    gnow = eng.g[side].clone()
    if fl == flags[0]:
        gref = gnow
    else:                                # kernel selections must agree (ablations will not)
        err = (gnow - gref).norm(dim=1) / gref.norm(dim=1).clamp_min(1e-30)
        print(f"    vs flags={flags[0]}: worst row {float(err.max()):.2e}, fro {float((gnow - gref).norm() / gref.norm()):.2e}, fails {int(eng.fail[0])}")
lib.wmf_debug_set_flags(0)
