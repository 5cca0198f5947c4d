"""tests/scale/fuzz_distributed.py at a fixed seed and a size that fits the CPU suite: AlsEngine on 2 .. 8 gloo ranks with the NumPy
stand-in kernels over drawn shapes, chunk counts, exchange modes, need lists and builds, against the single-process oracle."""
import os
import subprocess
import sys

from conftest import ROOT


def test_randomised_multi_rank_sweep():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scale", "fuzz_distributed.py"), "10", "31"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    tail = "\n".join(l for l in (res.stdout + res.stderr).splitlines() if not l.startswith("[Gloo]"))[-3000:]
    assert res.returncode == 0 and "10 cases in" in res.stdout, tail
