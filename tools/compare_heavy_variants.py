"""Full-scale cross-check of the heavy-row kernels (not a test): one half step of a bench configuration's side, solved by
every variant the debug flags select, ALL rows compared against the first (row-wise relative difference).  Catches what a
sample of rows cannot: a rare race between consecutive rows of one wave.
Usage: python tools/compare_heavy_variants.py cfg3 items 0,4096 [repeats]   (8192, the f32-MFMA accumulation, needs a -DWMF_LAB build)"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, '.')
from recmodel_amd import _lib, synth, WMF
from recmodel_amd.engine import AlsEngine, _ptr, _stream

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
side = sys.argv[2] if len(sys.argv) > 2 else "items"
flags = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,4096").split(",")]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
lib = _lib.load()
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda")
val = 10 * torch.log(1 + val)
eng = AlsEngine(n_users, n_items, k, bias, 0.1)
eng.set_interactions(ip, idx, val)
eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items)
eng.half_step("users"); eng.half_step("items"); eng.half_step("users")
fixed = eng._other(side)
eng.prepare(fixed)
c = eng.csr[side]
ref = None
for fl in flags:
    for rep in range(reps):
        lib.wmf_debug_set_flags(fl)
        eng.g[side].fill_(7.0)
        _lib.check(lib.wmf_solve_rows(c._plan, _ptr(eng.V[fixed]), _ptr(eng.bias_vec[fixed]) if bias else None, _ptr(c.indptr),
                                      _ptr(c.indices), _ptr(c.values), c.n_rows, eng.f, eng.ld, _ptr(eng.g[side]), _ptr(eng.fail), _stream()))
        torch.cuda.synchronize()
        g = eng.g[side][: eng.n_local[side], : eng.f].double()
        if ref is None:
            ref = g.clone()
            print(f"{cfg} {side}: {g.shape[0]} rows, reference = flags {fl}; fail count {int(eng.fail[0])}")
            continue
        num = (g - ref).norm(dim=1); den = ref.norm(dim=1).clamp_min(1e-30)
        rel = num / den
        print(f"flags={fl} rep={rep}: worst row {rel.max().item():.3e} (row {int(rel.argmax())}), rows above 1e-4: {int((rel > 1e-4).sum())}, "
              f"mean {rel.mean().item():.2e}, identical rows {int((num == 0).sum())}")
lib.wmf_debug_set_flags(0)
