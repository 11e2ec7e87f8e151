"""GPU box: randomized differential run of the fp64 handle against the fp64 C oracle across the configuration space
(variant x integrator x control mode x T x clipped x training x t_step x episode length), random actions incl.
saturating ones, full step with reward / done / solved / Euler rates / effort, several resets per case.
Prints one line per case and a summary; exits non-zero on any mismatch above 1e-8 (scaled) or any flag mismatch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from oracle import quad_oracle as O

O.build()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
SC = np.array([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0])
worst, bad = 0.0, 0
for case in range(n_cases):
    variant = ["main", "opt"][rng.randint(2)]
    integ = ["rk45", "dp5", "rk4"][rng.randint(3)]
    direct = int(rng.randint(2))
    T = int(rng.choice([1, 2, 5]))
    clipped = bool(rng.randint(2))
    training = bool(rng.randint(2))
    t_step = float(rng.choice([0.01, 0.005, 0.02]))
    n_max = int(rng.choice([8, 30, 1000]))
    N, S = 24, int(rng.choice([12, 40]))
    kw = dict(training=training, direct_control=direct, T=T, clipped=clipped, variant=variant, integrator=integ)
    env = aq.BatchedQuad(N, t_step, n_max, dtype=torch.float64, compute=({"lean": "lean"}.get(os.environ.get("QENV_FUZZ_HANDLE", "f64"))), **kw)
    refs = [O.OracleQuad(t_step, n_max, euler=0, **kw) for _ in range(N)]
    err, flags_ok = 0.0, True
    for ep in range(2):
        x0 = aq.random_init_states(rng, N, variant) * rng.choice([0.2, 1.0])
        x0[:, 6:10] /= np.linalg.norm(x0[:, 6:10], axis=1, keepdims=True)
        st, ac = env.reset(x0.T.copy())
        for e, r in enumerate(refs):
            s_ref, _ = r.reset(x0[e])
            err = max(err, np.max(np.abs(st[:, :13, e].cpu().numpy() - s_ref[:, :13]) / np.maximum(np.abs(s_ref[:, :13]), 1e-3 * SC)))
        for k in range(S):
            if direct:
                a = rng.normal(0, rng.choice([0.3, 1.5]), (4, N))
            else:
                a = np.array([1.03 * 9.82, 0, 0, 0])[:, None] + rng.normal(0, 1, (4, N)) * np.array([rng.choice([1, 15]), 0.3, 0.3, 0.05])[:, None]
            obs, rew, done = env.step(a)
            S_, R_, D_ = env.state.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy()
            AV, EF, SO = env.ang_vel.cpu().numpy(), env.step_effort.cpu().numpy(), env.solved.cpu().numpy()
            for e, r in enumerate(refs):
                _, rr, dd = r.step(a[:, e])
                err = max(err, np.max(np.abs(S_[:, e] - r.state) / np.maximum(np.abs(r.state), 1e-3 * SC)))
                err = max(err, abs(R_[e] - rr) / max(1.0, abs(rr)))
                err = max(err, np.max(np.abs(AV[:, e] - r.ang_vel)) / 100.0, np.max(np.abs(EF[:, e] - r.step_effort)) / max(1.0, np.abs(r.step_effort).max()))
                if bool(D_[e]) != bool(dd) or int(SO[e]) != int(r.solved):
                    flags_ok = False
    ok = err < 1e-8 and flags_ok
    worst = max(worst, err); bad += 0 if ok else 1
    print(json.dumps(dict(case=case, variant=variant, integ=integ, direct=direct, T=T, clipped=clipped, training=training,
                          t_step=t_step, n_max=n_max, err=float(err), flags_ok=flags_ok, ok=ok)), flush=True)
print(json.dumps(dict(handle=os.environ.get("QENV_FUZZ_HANDLE", "f64"), cases=n_cases, mismatching=bad, worst_scaled_err=float(worst))))
sys.exit(1 if bad else 0)
