"""CPU (the oracle, scipy-exact): config 4's census -- the reference's trained N = 128 actor in the loop (deterministic mean
action, T = 5 window, 500 steps from config 4's own initial states): RK45 sub-steps per env step, per lane and per wave of 64
(the env step of k_ppo_fused runs 64 envs on one wave), natural order and sorted by candidate keys.
Round 5 (1 024 envs): lanes 1.015, waves 1.47 in the natural order, 1.35 sorted by the episode's totals or by |omega| at its
end -- 64 % of the extra sub-steps fall into the first 100 steps (2.1 per wave-step there: every env's transient, at random
steps), only 1 % of the envs stay hard: a static order can save ~0.1 x 4.9 k cycles of a 51 k block-step, which is what the
GPU measured (2.80 against 2.85 G env-steps/s: nothing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import quad_oracle as O
from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
O.build()
N, S, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 500, 5
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "g9_mlp_N128.npz"), allow_pickle=False)
A = [np.asarray(g[f"actor_{l}_{p}"], dtype=np.float64) for l in (0, 2, 4) for p in ("weight", "bias")]
x0 = fast_random_init_states(3, 131072, 0, N)
nsub = np.zeros((N, S), dtype=np.int16); fin = np.zeros((N, 13))
t0 = time.time()
for e in range(N):
    r = O.OracleQuad(0.01, 500, training=False, direct_control=1, T=T)
    s, a = r.reset(x0[e])
    win = O.OracleWindow(T)
    obs = win.dl_input(s, a)
    for k in range(S):
        act = O.mlp_forward(obs.astype(np.float64), *A, True)[0]
        qs, rew, d = r.step(act)
        nsub[e, k] = r.nsub
        obs = win.dl_input(qs, act[None, :])
    fin[e] = r.state
print("oracle time %.1f s" % (time.time() - t0))
def wave_cost(order, sl=slice(None)):
    return nsub[order][:, sl].reshape(N // 64, 64, -1).max(axis=1).mean()
print("lane-level mean sub-steps per step %.4f; per 50-step phase:" % nsub.mean(), np.round(nsub.reshape(N, 10, 50).mean(axis=(0, 2)), 3))
print("wave-level, natural order %.3f; per phase:" % wave_cost(np.arange(N)), np.round(nsub.reshape(N // 64, 64, 10, 50).max(axis=1).mean(axis=(0, 2)), 3))
tot = nsub.sum(1)
print("sorted by the episode's totals %.3f" % wave_cost(np.argsort(-tot, kind="stable")))
for name, key in (("|omega| at the end", np.linalg.norm(fin[:, 10:13], axis=1)), ("|v| at the end", np.linalg.norm(fin[:, [1, 3, 5]], axis=1)),
                  ("|pos| at the end", np.linalg.norm(fin[:, [0, 2, 4]], axis=1))):
    print("sorted by %s: %.3f (corr with totals %.2f)" % (name, wave_cost(np.argsort(-key, kind="stable")), np.corrcoef(key, tot)[0, 1]))
print("share of envs with > 10 %% multi-sub-step steps: %.3f; share of multi-sub-step steps in the first 100 steps: %.3f" %
      (((nsub > 1).mean(axis=1) > 0.1).mean(), (nsub[:, :100] > 1).sum() / max(1, (nsub > 1).sum())))
