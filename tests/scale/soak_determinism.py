"""Soak (GPU, hand-run): T ALS iterations of a bench configuration from the same initial item factors, R times; every repetition must
end in bit-identical user and item factors (T x (n_users + n_items) row solves per repetition through the resident-loop path of
bench.py).  Usage: python tests/scale/soak_determinism.py [cfg3] [T=10] [R=3] [zipf=0]"""
import sys, time
sys.path.insert(0, '.')
import torch
from recmodel_amd import WMF, synth
from recmodel_amd.engine import AlsEngine

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
zipf = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda", zipf_a=zipf)
eng = AlsEngine(n_users, n_items, k, bias, 0.1)
eng.set_interactions(ip, idx, 10 * torch.log(1 + val))
del ip, idx, val
items0 = WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items
ref = None
for rep in range(R):
    eng.set_factors("items", items0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(T):
        eng.half_step("users"); eng.half_step("items")
    eng.check_numerics()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    got = (eng.factors["users"].clone(), eng.factors["items"].clone())
    if ref is None:
        ref = got
        print(f"{cfg} zipf={zipf}: repetition 0: {T} iterations in {dt * 1e3:.0f} ms")
    else:
        du = int((ref[0] != got[0]).any(dim=1).sum().item()); di = int((ref[1] != got[1]).any(dim=1).sum().item())
        print(f"repetition {rep}: user rows that differ {du}, item rows that differ {di}  ({dt * 1e3:.0f} ms)")
        assert du == 0 and di == 0
print("bit-identical")
