import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def csr_from(g, prefix):
    shape = tuple(int(x) for x in g[f"{prefix}_shape"])
    return sp.csr_matrix((g[f"{prefix}_data"], g[f"{prefix}_indices"], g[f"{prefix}_indptr"]), shape=shape)


@pytest.fixture(scope="session")
def golden():
    return load_golden
