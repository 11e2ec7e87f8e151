"""CPU (oracle, fp64): RK45 attempts per (env, step) of config 3's closed loop (LQR, direct_control = 0, 500-step
episodes from random initial states) -- how a 64-env wave's cost (max over its lanes, per step) compares with the lanes'
own need, and what regrouping the envs by the previous chunk's attempt count would recover.
python tests/tools/attempt_census.py [envs] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import quad_oracle as qo

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.RandomState(0)
init = qo.random_init_states(rng, n_envs)
att = np.zeros((steps, n_envs), np.int32)
alive = np.ones((steps, n_envs), bool)
for e in range(n_envs):
    env = qo.OracleQuad(0.01, steps, training=True, direct_control=0, T=1)
    env.reset(init[e])
    lqr = qo.OracleLqr(env)
    for s in range(steps):
        env.step(lqr.control())
        att[s, e] = (env.nfev - 2) // 6
        if env.done:
            alive[s + 1:, e] = False   # the device freezes a finished env (its lane idles)
            break
att = att * alive
print("envs", n_envs, "steps", steps, "alive fraction", alive.mean())
print("attempts per live env-step: mean %.3f" % (att.sum() / alive.sum()))
for lo in range(0, steps, 50):
    blk = att[lo:lo + 50]; al = alive[lo:lo + 50]
    print("steps %3d..%3d: lane mean %.3f  per-env mean p50 %.2f p90 %.2f p99 %.2f" % (
        lo, lo + 49, blk.sum() / max(al.sum(), 1), *np.percentile(blk.mean(0), [50, 90, 99])))

def wave_cost(order_fn, chunk):
    """sum over chunks and waves of sum_s max_lane attempts, the envs regrouped at every chunk start by order_fn(previous chunk)"""
    tot = 0
    prev = None
    for lo in range(0, steps, chunk):
        blk = att[lo:lo + chunk]
        order = np.arange(n_envs) if prev is None else order_fn(prev)
        w = blk[:, order].reshape(blk.shape[0], -1, 64).max(2)
        tot += w.sum()
        prev = blk
    return tot / (steps * n_envs / 64)

print("wave attempts per wave-step, in env order (any chunking): %.3f" % wave_cost(lambda p: np.arange(n_envs), steps))
for chunk in (250, 100, 50, 25, 10):
    print("chunk %3d: sorted by previous chunk's attempts %.3f   2-class partition (sum > chunk) %.3f" % (
        chunk, wave_cost(lambda p: np.argsort(p.sum(0), kind="stable"), chunk),
        wave_cost(lambda p: np.argsort(p.sum(0) > p.shape[0], kind="stable"), chunk)))
