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


def free_port():
    """A TCP port for a rendezvous that starts a few seconds from now.  Drawn OUTSIDE the kernel's ephemeral range: a
    port handed out by bind(("", 0)) comes from that range and can be given to somebody's outgoing connection before rank
    0 listens on it (EADDRINUSE once in ~500 spawns of the fuzzers -- a 4 % flake per run of this suite)."""
    import random
    import socket
    lo, hi = 20000, 32000
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            hi = max(lo + 1000, min(hi, int(f.read().split()[0]) - 1))
    except (OSError, ValueError, IndexError):
        pass
    rnd = random.SystemRandom()  # (never the seeded global generator of a test)
    for _ in range(64):
        p = rnd.randrange(lo, hi)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", p))
                return p
            except OSError:
                continue
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


# Every error a parity test computes through rel_err / elem_rel_err is kept per test and written to
# gpurun_out/parity_errors.json at the end of a session that ran on a GPU (copied to profiles/rNN/): the measured
# distance to the oracle, not just "below the threshold".
_CURRENT = [None]
_ERRORS = {}


@pytest.fixture(autouse=True)
def _parity_log_current_test(request):
    _CURRENT[0] = request.node.nodeid
    yield
    _CURRENT[0] = None


def _log_err(kind, value, size):
    if _CURRENT[0] is None:
        return
    rec = _ERRORS.setdefault(_CURRENT[0], {})
    k = rec.setdefault(kind, {"comparisons": 0, "max": 0.0, "largest_array": 0})
    k["comparisons"] += 1
    k["max"] = max(k["max"], float(value)) if np.isfinite(value) else float("nan")
    k["largest_array"] = max(k["largest_array"], int(size))


def pytest_sessionfinish(session, exitstatus):
    try:
        import torch
        if not _ERRORS or not torch.cuda.is_available():
            return
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_errors.json")
        tests = {}
        try:  # several pytest invocations of one box session add up (a later run of a test replaces its entry)
            tests = json.load(open(path)).get("tests", {})
        except Exception:
            pass
        tests.update(_ERRORS)
        doc = {"what": "largest error each GPU parity test measured against the oracle (rel_err: max|a-b| / max|b|; "
                       "elem_rel_err: element-wise with a floor of 1e-3 x the largest entry unless the test says otherwise)",
               "device": torch.cuda.get_device_name(0), "exitstatus": int(exitstatus), "tests": tests}
        with open(path, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    except Exception:
        pass


def rel_err(a, b):
    """max |a - b| / max(|b|_inf, tiny): the relative error the 1e-5 tolerance of north_star is stated in."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    e = float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
    _log_err("rel_err", e, a.size)
    return e


def elem_rel_err(a, b, floor_frac=1e-3):
    """ELEMENT-WISE relative error with an absolute floor: max_i |a_i - b_i| / max(|b_i|, floor_frac * max|b|).
    rel_err above is norm-wise (one large entry hides bad small ones); this one bounds every entry whose magnitude is
    at least floor_frac of the largest, and treats smaller entries as if they had that magnitude."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    floor = max(floor_frac * float(np.max(np.abs(b))), 1e-30)
    e = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
    _log_err("elem_rel_err(floor %g)" % floor_frac, e, a.size)
    return e
