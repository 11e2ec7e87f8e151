"""GPU box: the overlapped PPO rollout step (qenv_ppo_step: reset in the step kernel's epilogue, warm-up on a side stream beside
the next policy step) against the sequential calls (policy_act, step, reset_sampled): every rollout buffer, the env state
after finish() and the banked returns must be bit-identical; and the wall time per step of both."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from autonomous_quadrotor_environment_amd import ActorCriticWeights, PolicyRollout

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = int(sys.argv[2]) if len(sys.argv) > 2 else 96
handle = sys.argv[3] if len(sys.argv) > 3 else "lean"
res = {}
for overlap in (True, False):
    env = bench.build_env("config5", handle, "rk45", n, 0, n, dev)
    w = ActorCriticWeights.random(128, 75, 4, device=dev, seed=1); w.std = 0.1
    ro = PolicyRollout(env, w, H, store_obs=True, auto_reset=True, seed=4, env_offset=0)
    ro.overlap = overlap
    outs = []
    for rep in range(3):   # three horizons: resets pending across finish() boundaries are exercised too
        ro.t = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(H):
            ro.step()
        ret, adv = ro.finish(ddof=0, group=False, deterministic=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        outs.append([x.clone() for x in (ro.actions, ro.logprobs, ro.rewards, ro.masks, ro.obs, ro.values, ret, adv, env.state,
                                         env.last_return, env.episodes, env.window, env.i, env.flags)])
    res[overlap] = (outs, dt / H * 1e6)
names = "actions logprobs rewards masks obs values returns adv state last_return episodes window i flags".split()
bad = []
for rep in range(3):
    for nm, a, b in zip(names, res[True][0][rep], res[False][0][rep]):
        if not torch.equal(a, b):
            bad.append((rep, nm, float((a.double() - b.double()).abs().max())))
print(json.dumps(dict(envs=n, horizon=H, handle=handle, mismatches=bad, episodes_finished=int(res[True][0][2][10].sum()),
                      us_per_step_overlap=res[True][1], us_per_step_sequential=res[False][1])))
sys.exit(1 if bad else 0)
