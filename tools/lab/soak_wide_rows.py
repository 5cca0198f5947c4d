"""Lab: the same heavy-row half step R times at width f; counts the rows whose bits differ from the first run.
Usage: [WMF_HIP_LIB=...] python tools/lab/soak_wide_rows.py f [rows] [R]"""
import sys
sys.path.insert(0, '.')
import torch
from recmodel_amd import WMF, synth
from recmodel_amd.engine import AlsEngine
f = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000; R = int(sys.argv[3]) if len(sys.argv) > 3 else 50
ip, idx, val = synth.make_counts(n, 4000, 300, 7, device="cuda")
eng = AlsEngine(n, 4000, f - 1, True, 0.1)
eng.set_interactions(ip, idx, 10 * torch.log(1 + val))
eng.set_factors("items", WMF(num_items=4000, num_users=1, dim=f - 1, gamma=0.1, weighted=True, bias=True).items)
eng.half_step("users")
ref = eng.factors["users"].clone()
bad = 0
for _ in range(R):
    eng.half_step("users")
    bad += int((ref != eng.factors["users"]).any(dim=1).sum().item())
eng.check_numerics()
print(f"f={f}: {R} repetitions x {n} heavy rows: {bad} rows differed from the first run")
