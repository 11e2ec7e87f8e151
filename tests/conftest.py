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


def step_flag_margin(state, ang, variant="main", training=True):
    """Relative distance of each (fp64 reference) sample to the NEAREST decision threshold of done_condition /
    reward_function (quadrotor_env.py:500-573, quadrotor_env_opt.py:410-484): the nine bounding-box tests, the three
    reward tiers with their attitude sub-tests, and the solved box.  A handle whose state error is e can only take
    another branch than the reference on samples whose margin is below ~e / threshold; everywhere else integer
    outputs (done, solved) and the branch-dependent reward must agree exactly.  state [n, 13], ang [n, 3]."""
    state, ang = np.atleast_2d(np.asarray(state, dtype=np.float64)), np.atleast_2d(np.asarray(ang, dtype=np.float64))
    v, w = state[:, 1:6:2], state[:, 10:13]
    if variant == "opt":
        bb = np.array([10, 10, 10, np.pi / 2, np.pi / 2, 4, 30, 30, 30.0]) * (1 if training else 1000)
        tr, att_k = (0.001, 0.01, 0.1), 3.4641016151377544
        att = np.linalg.norm(ang, axis=1)
        target = 9 * 0.001 ** 2
    else:
        bb = np.array([10, 10, 10, np.pi / 2, np.pi / 2, 0.75 * np.pi, 20, 20, 20.0])
        tr, att_k = (0.005, 0.01, 0.1), 5.656854249492381
        att = np.linalg.norm(ang[:, :2], axis=1)
        target = 9 * 0.005 ** 2
    c = np.abs(np.concatenate([v, ang, w], 1))
    m = np.min(np.abs(c - bb) / bb, axis=1)
    r_norm = np.sqrt(np.sum(v * v, 1) + ang[:, 2] ** 2)
    for t in tr:
        m = np.minimum(m, np.abs(r_norm - 2 * t) / (2 * t))
        m = np.minimum(m, np.abs(att - att_k * t) / (att_k * t))
    current = np.sum(v * v, 1) + np.sum(ang * ang, 1) + np.sum(w * w, 1)
    return np.minimum(m, np.abs(current - target) / target)


def visual_reward_margin(rows):
    """The same for visual_reward (rl_reward_fuction.py:3-102) on the g11 / fuzz row layout: marker 0:3, pos 3:6,
    vel 6:9, ang 14:17, v_ang 17:20."""
    rows = np.asarray(rows, dtype=np.float64)
    mk, pos, vel, ang, wa = rows[:, 0:3], rows[:, 3:6], rows[:, 6:9], rows[:, 14:17], rows[:, 17:20]
    exy = np.linalg.norm(mk[:, :2] - pos[:, :2], axis=1)
    ez = np.abs(mk[:, 2] - 5 - pos[:, 2])
    rel = lambda x, t: np.abs(x - t) / np.maximum(np.abs(t), 1e-2)
    m = rel(np.linalg.norm(vel, axis=1), 0.5196152422706632)
    for x, t in ((pos[:, 2], -4.95), (exy, 0.14), (np.linalg.norm(ang[:, :2], axis=1), 0.4937019546242588),
                 (np.linalg.norm(wa, axis=1), 1.4142135623730951), (ez, 7.0)):
        m = np.minimum(m, rel(x, t))
    return np.minimum(m, rel(exy, ez / 7 * 5 + 0.2))
