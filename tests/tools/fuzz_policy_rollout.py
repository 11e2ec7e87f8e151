"""GPU box: the PPO worker loop (policy step -> masked auto-reset with device-drawn states -> T warm-up steps ->
window) in fp64 against the oracle, teacher-forced with the device's own actions and reset states (the MLP has its own
tests): rewards, masks, the 75-float observation of every step, final states, banked returns and episode counts over
random (variant, T = 5 window, training, episode length, integrator) points."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from oracle import quad_oracle as O

O.build()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SC = np.array([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0])
worst, bad = 0.0, 0
for case in range(n_cases):
    variant = ["main", "opt"][rng.randint(2)]
    integ = ["rk45", "dp5", "rk4"][rng.randint(3)]
    training = bool(rng.randint(2))
    n_max = int(rng.choice([6, 15, 60]))
    hidden = int(rng.choice([32, 128]))
    N, H, T = 16, int(rng.choice([30, 70])), 5
    kw = dict(training=training, direct_control=1, T=T, variant=variant, integrator=integ)
    w = aq.ActorCriticWeights.random(hidden, 75, 4, seed=int(rng.randint(1 << 30)))
    w.std = float(rng.choice([0.1, 0.6]))  # 0.6: saturating actions, envs leave the box
    env = aq.BatchedQuad(N, 0.01, n_max, dtype=torch.float64, compute=({"lean": "lean"}.get(os.environ.get("QENV_FUZZ_HANDLE", "f64"))), **kw)
    ro = aq.PolicyRollout(env, w, H, store_obs=True, auto_reset=True, seed=int(rng.randint(1 << 30)), env_offset=int(rng.randint(1000)))
    ro.keep_init = True
    ro.start()
    init0 = ro.init.cpu().numpy().copy()
    inits, dones = [], []
    for k in range(H):
        ro.step()
        inits.append(ro.init.cpu().numpy().copy()); dones.append(ro.done.cpu().numpy().copy())
    A, R, M, OBS = (x.cpu().numpy() for x in (ro.actions, ro.rewards, ro.masks, ro.obs))
    S_, last, eps = env.state.cpu().numpy(), env.last_return.cpu().numpy(), env.episodes.cpu().numpy()
    err, flags_ok = 0.0, True
    for e in range(N):
        r = O.OracleQuad(0.01, n_max, euler=0, **kw)
        win = O.OracleWindow(T)
        s, a = r.reset(init0[:, e]); obs = win.dl_input(s, a).copy()
        ep_ret, n_ep, banked = 0.0, 0, 0.0
        for k in range(H):
            err = max(err, float(np.max(np.abs(OBS[k, :, e] - obs))))
            qs, rew, d = r.step(A[k, :, e])
            ep_ret += rew
            err = max(err, abs(R[k, e] - rew) / max(1.0, abs(rew)))
            flags_ok &= (M[k, e] == (0.0 if d else 1.0)) and (bool(dones[k][e]) == d)
            obs = win.dl_input(qs, A[k, :, e][None, :]).copy()
            if d:
                banked, ep_ret, n_ep = ep_ret, 0.0, n_ep + 1
                win.reset()
                s, a = r.reset(inits[k][:, e]); obs = win.dl_input(s, a).copy()
        err = max(err, float(np.max(np.abs(S_[:, e] - r.state) / np.maximum(np.abs(r.state), 1e-3 * SC))))
        flags_ok &= int(eps[e]) == n_ep and (n_ep == 0 or abs(last[e] - banked) < 1e-9 * max(1.0, abs(banked)))
    ok = flags_ok and err < 1e-6  # observations are float32: 6e-8 relative on values up to ~10
    worst = max(worst, err); bad += 0 if ok else 1
    print(json.dumps(dict(case=case, variant=variant, integ=integ, training=training, n_max=n_max, hidden=hidden, H=H, std=w.std,
                          resets=int(eps.sum()), err=float(err), flags_ok=bool(flags_ok), ok=bool(ok))), flush=True)
print(json.dumps(dict(handle=os.environ.get("QENV_FUZZ_HANDLE", "f64"), cases=n_cases, mismatching=bad, worst_err=float(worst))))
sys.exit(1 if bad else 0)
