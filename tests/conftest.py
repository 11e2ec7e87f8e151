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


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import quad_oracle
    quad_oracle.build()
    return quad_oracle


def rel_err(a, b, floor=1e-3):
    """max_k |a_k-b_k| / max(|b_k|, floor): the metric of SURVEY.md App. C E3."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


# bounding-box scales of the 13 state components (pos 5 m, vel 10 m/s, quaternion 1, body rate 20 rad/s:
# quadrotor_env.py:30-34,139-143); SURVEY.md H2's "abs floor 1e-3*scale"
STATE_SCALE = np.array([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0])


def scaled_err(a, b, scale=STATE_SCALE):
    """max_k |a_k-b_k| / max(|b_k|, 1e-3*scale_k) over trailing dimension 13."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * scale)))
