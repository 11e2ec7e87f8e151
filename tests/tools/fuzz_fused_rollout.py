"""GPU box: the fused rollout launch (k_ppo_fused: qenv_ppo_rollout in-band / qenv_policy_rollout as ONE launch) against the
per-step launches (PolicyRollout.step: k_mlp_v3 / k_mlp_small + k_step) over random points: handle type, batch size (ragged
blocks and tiles), episode length (restarts inside the launch), policy std (saturating actions), horizon and the way it is cut
into library calls (ring head != 0 at a call's start), stored observations as windows / slices / not at all, injected noise
or Philox, sampling or the deterministic evaluation loop.  Every rollout buffer and every env field bit for bit.

    python tests/tools/fuzz_fused_rollout.py [seed] [cases]
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
HANDLES = {"lean": dict(dtype=torch.float64, compute="lean"), "f32": dict(dtype=torch.float32),
           "mixed": dict(dtype=torch.float64, compute="f32"), "f64": dict(dtype=torch.float64)}
FIELDS = ("state", "window", "i", "ang", "ang_vel", "reward", "episode_return", "last_return", "episodes", "done", "solved")
bad = 0
for case in range(n_cases):
    handle = list(HANDLES)[rng.randint(4)]
    kw = HANDLES[handle]
    n = int(rng.choice([1, 31, 64, 65, 200, 1000, 2129, 4160]))
    H = int(rng.choice([7, 23, 40]))
    n_max = int(rng.choice([6, 13, 60, 500]))
    sampling = bool(rng.randint(2))
    cuts = sorted(set(int(c) for c in rng.randint(1, H, size=rng.randint(0, 3))))
    w = aq.ActorCriticWeights.random(128, 75, 4, seed=int(rng.randint(1 << 30)))
    w.std = float(rng.choice([0.1, 0.6]))
    seed, off = int(rng.randint(1 << 30)), int(rng.randint(5000))
    store = bool(rng.randint(2)) if sampling else False
    slices = bool(rng.randint(2)) if store else None
    inject = sampling and bool(rng.randint(2))
    block = [None, 32, 64][rng.randint(3)]   # env block per wavefront of the fused launch: chosen by the batch size / forced
    dyn_first = bool(rng.randint(4) == 0)   # a dynamics-only step in front: the launch starts with stale prev_ang rows (ang_ok = 0)
    res = []
    for mode in ("step", "run"):
        env = aq.BatchedQuad(n, 0.01, n_max, training=bool(case & 1), direct_control=1, T=5, **kw)
        env.set_ppo_fused(mode != "step", block=block)
        if sampling:
            ww = w if store else aq.ActorCriticWeights(w.actor, None, env.device)   # (no stored observations: no batched critic)
            ww.std = w.std
            ro = aq.PolicyRollout(env, ww, H, store_obs=store, auto_reset=True, seed=seed, env_offset=off, inband=True,
                                  slice_obs=slices)
        else:
            ro = aq.PolicyRollout(env, aq.ActorCriticWeights(w.actor, None, env.device), H)
        ro.start(None if sampling else env.sample_init_states(seed, 0, off))
        if dyn_first:
            env.step(None, flags=L.STEP_DYNAMICS_ONLY)
        noise = None
        if inject:
            g = torch.Generator(device="cpu"); g.manual_seed(seed)
            noise = torch.randn(H, 4, n, generator=g, dtype=torch.float64).to(env.device, env.dtype)
        if mode == "step":
            for t in range(H):
                ro.step(noise=None if noise is None else noise[t]) if sampling else ro.step(deterministic=True)
        else:
            edges = [0] + cuts + [H]
            for a, b in zip(edges[:-1], edges[1:]):
                if sampling:
                    ro.run(b - a, noise=None if noise is None else noise[a:b])
                else:
                    ro.run_deterministic(b - a)
        bufs = [ro.actions, ro.rewards, ro.masks, ro.done]
        if sampling:
            bufs += [ro.logprobs, ro.valid] + ([ro.obs] if store else [])
        env.ppo_flush()
        res.append([x.clone() for x in bufs] + [getattr(env, f).clone() for f in FIELDS])
    diff = [k for k, (x, y) in enumerate(zip(*res)) if not torch.equal(x, y)]
    ok = not diff
    bad += 0 if ok else 1
    print(json.dumps(dict(case=case, handle=handle, n=n, H=H, cuts=cuts, n_max=n_max, sampling=sampling, std=w.std, store=store,
                          slices=slices, inject=inject, dyn_first=dyn_first, block=block, restarts=int(res[0][-3].sum()), mismatching_buffers=diff, ok=ok)), flush=True)
print(json.dumps(dict(cases=n_cases, mismatching=bad)))
sys.exit(1 if bad else 0)
