import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not p.endswith("kernels.npz"))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def max_rel(a, b):
    """max |a-b| / max |b| : vector-norm relative error scaled by the field maximum."""
    d = float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()) if a.size else 0.0
    s = float(np.abs(b).max()) if b.size else 0.0
    return d / s if s > 0 else d


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def params():
    from sph_amd.params import default_params
    return default_params()
