"""The randomised sweeps of tests/scale/ at a fixed seed and a size that fits the regular GPU run (each is a script of its own:
one child process per sweep, one at a time).  DESIGN.md section 2 lists what each one draws and what it has found."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,args", [("fuzz_parity.py", ["60", "21"]), ("fuzz_parity.py", ["40", "22", "145-260"]),
                                         ("fuzz_train.py", ["80", "23"]), ("fuzz_csr.py", ["60", "24"]), ("fuzz_topn.py", ["60", "25"]),
                                         ("fuzz_f64.py", ["20", "26"])])
def test_randomised_sweep(script, args):
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scale", script), *args], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=900)
    tail = "\n".join((res.stdout + res.stderr).splitlines()[-15:])
    assert res.returncode == 0, tail
