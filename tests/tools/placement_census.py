"""CPU (the oracle, scipy-exact): how many RK45 sub-steps does every env of config 3 take per env step over a 500-step
LQR episode, and what does a WAVE of 64 lanes pay -- the maximum over its lanes -- in the batch's natural order and with the
envs sorted by hardness?  2 048 envs of the bench's own initial states (fast_random_init_states(3, ...)).  Round 5:
lane-level mean 1.019 sub-steps per step, wave-level 1.73 in the natural order, 1.24 sorted by the episode's total (1.25
sorted by |omega| at the end of the episode: what BatchedQuad.regroup() uses); 5 % of the envs take > 1 sub-step in more
than 10 % of their steps (the ones the LQR never stabilises)."""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import quad_oracle as O
from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
O.build()
N, S = 2048, 500
x0 = fast_random_init_states(3, 262144, 0, N)
kw = dict(training=False, direct_control=0, T=1, clipped=True, variant="main", integrator="rk45")
nsub = np.zeros((N, S), dtype=np.int16)
t0 = time.time()
for e in range(N):
    r = O.OracleQuad(0.01, 1000, euler=0, **kw); r.reset(x0[e]); c = O.OracleLqr(r, True)
    for k in range(S):
        r.step(c.control())
        nsub[e, k] = r.nsub
print("oracle time", time.time() - t0)
att = nsub  # attempts ~ accepted sub-steps (+ rejected ones not counted here)
print("lane-level mean sub-steps per step", att.mean())
def wave_cost(order):
    a = att[order].reshape(N // 64, 64, S)
    return a.max(axis=1).mean()
nat = wave_cost(np.arange(N))
tot = att.sum(axis=1)
srt = wave_cost(np.argsort(tot))
print("wave-level mean (natural order)", nat, " sorted by episode hardness", srt)
# by phases of the episode (20-step regions)
a = att.reshape(N // 64, 64, S).max(axis=1)   # [waves, S]
print("natural, per 50-step phase:", np.round(a.reshape(-1, 10, 50).mean(axis=(0, 2)), 3))
a2 = att[np.argsort(tot)].reshape(N // 64, 64, S).max(axis=1)
print("sorted,  per 50-step phase:", np.round(a2.reshape(-1, 10, 50).mean(axis=(0, 2)), 3))
print("share of envs with any multi-sub-step step:", (att.max(axis=1) > 1).mean(), " envs with > 10 % multi steps:", ((att > 1).mean(axis=1) > 0.1).mean())
# hardness known only from the FIRST 20 steps
early = att[:, :20].sum(axis=1)
print("sorted by the first 20 steps' sub-steps:", wave_cost(np.argsort(early)))
