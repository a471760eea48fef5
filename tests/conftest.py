import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def rel_err(a, b):
    """max |a - b| / max(|b|_inf, tiny): the relative error the 1e-5 tolerance of north_star is stated in."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def elem_rel_err(a, b, floor_frac=1e-3):
    """ELEMENT-WISE relative error with an absolute floor: max_i |a_i - b_i| / max(|b_i|, floor_frac * max|b|).
    rel_err above is norm-wise (one large entry hides bad small ones); this one bounds every entry whose magnitude is
    at least floor_frac of the largest, and treats smaller entries as if they had that magnitude."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    floor = max(floor_frac * float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
