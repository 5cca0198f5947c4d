"""Lab: kernel time of the four-waves-per-row kernel on heavy rows at a given width (through the engine's profile table).
Usage: [WMF_HIP_LIB=...] python tools/lab/time_wide_rows.py f [rows]"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from recmodel_amd import _lib, synth
from recmodel_amd.engine import AlsEngine
f = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
lib = _lib.load()
ip, idx, val = synth.make_counts(n, 4000, 300, 7, device="cuda")
eng = AlsEngine(n, 4000, f - 1, True, 0.1)
eng.set_interactions(ip, idx, 10 * torch.log(1 + val))
from recmodel_amd import WMF
eng.set_factors("items", WMF(num_items=4000, num_users=1, dim=f - 1, gamma=0.1, weighted=True, bias=True).items)
for _ in range(2): eng.half_step("users")
lib.wmf_profile_reset(); lib.wmf_profile_enable(1)
for _ in range(5): eng.half_step("users")
torch.cuda.synchronize(); lib.wmf_profile_enable(0)
for nm, tag, ms, cnt, lo, hi in sorted(_lib.profile_table(lib), key=lambda e: -e[2])[:2]:
    print(f"f={f} {nm}: {ms / cnt:.3f} ms per launch ({n} rows of ~300 entries)")
