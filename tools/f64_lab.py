"""Timing lab (GPU, hand-run): per-kernel times of the float64 half steps (wmf_half_step_f64) on a synth.CONFIGS workload.
Usage: python tools/f64_lab.py [cfg2]"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from recmodel_amd import _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
lib = _lib.load()
if len(sys.argv) > 2:
    lib.wmf_debug_set_flags(int(sys.argv[2]))
dev = torch.device("cuda:0")
lib.wmf_profile_reset()
lib.wmf_profile_enable(1)
out = bench.float64_path(cfg, dev, lib, iters=2)
lib.wmf_profile_enable(0)
print(out)
for nm, tag, ms, n, lo, hi in sorted(_lib.profile_table(lib), key=lambda e: -e[2]):
    print(f"{nm:40s} launches={n:3d} avg={ms / n:8.3f} ms  min={lo:8.3f} max={hi:8.3f}")
