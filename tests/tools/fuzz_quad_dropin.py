"""GPU box: the N = 1 drop-in class `quad` (numpy in / numpy out, the reference's attribute surface) against the oracle
over random API sequences: seed() + reset() with the reference's RNG draw order, reset(det_state), steps with random
(also saturating) actions, stepping on after done, both variants, both control modes, T in {1, 2, 5}: returned
(state, reward, done) and the attributes controllers read afterwards."""
import os, sys, json, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import autonomous_quadrotor_environment_amd as aq
from oracle import quad_oracle as O

O.build()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SC = np.array([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0])
worst, bad = 0.0, 0
for case in range(n_cases):
    variant = ["main", "opt"][rng.randint(2)]
    direct = int(rng.randint(2)); T = int(rng.choice([1, 2, 5])); training = bool(rng.randint(2))
    clipped = True if variant == "opt" else bool(rng.randint(2))
    n_max = int(rng.choice([6, 25, 1000]))
    with contextlib.redirect_stdout(io.StringIO()):  # the class prints 'Environment Condition: ...' like the reference
        q = aq.quad(0.01, n_max, training=training, euler=0, direct_control=direct, T=T, clipped=clipped, variant=variant)
    r = O.OracleQuad(0.01, n_max, training=training, euler=0, direct_control=direct, T=T, clipped=clipped, variant=variant)
    err, ok_flags = 0.0, True

    def cmp(tag):
        global err, ok_flags
        e = float(np.max(np.abs(np.asarray(q.state).ravel() - r.state) / np.maximum(np.abs(r.state), 1e-3 * SC)))
        e = max(e, float(np.max(np.abs(np.asarray(q.ang).ravel() - r.ang))), abs(float(q.abs_sum) - float(r.abs_sum)) / max(1.0, abs(r.abs_sum)))
        err = max(err, e)
        ok_flags &= bool(q.done) == bool(r.done) and int(q.solved) == int(r.solved) and int(q.i) == int(r.i)

    for ep in range(3):
        if rng.randint(2):
            seed = int(rng.randint(1 << 30))
            q.seed(seed); r.seed(seed)
            s1, a1 = q.reset(); s2, a2 = r.reset()
        else:
            x0 = aq.random_init_states(rng, 1, variant)[0] * rng.choice([0.1, 1.0])
            x0[6:10] /= np.linalg.norm(x0[6:10])
            s1, a1 = q.reset(x0); s2, a2 = r.reset(x0)
        err = max(err, float(np.max(np.abs(np.asarray(s1) - s2))), float(np.max(np.abs(np.asarray(a1) - a2))))
        cmp("reset")
        for k in range(int(rng.choice([3, 12, 30]))):
            a = rng.normal(0, rng.choice([0.3, 2.0]), 4) if direct else np.array([1.03 * 9.82, 0, 0, 0]) + rng.normal(0, 1, 4) * np.array([rng.choice([1, 10]), 0.2, 0.2, 0.05])
            o1, rw1, d1 = q.step(a); o2, rw2, d2 = r.step(a)
            err = max(err, float(np.max(np.abs(np.asarray(o1).ravel() - np.asarray(o2).ravel()) / np.maximum(np.abs(np.asarray(o2).ravel()), 1e-3))),
                      abs(float(rw1) - float(rw2)) / max(1.0, abs(float(rw2))))
            ok_flags &= bool(d1) == bool(d2)
            cmp("step")
    ok = ok_flags and err < 1e-8
    worst = max(worst, err); bad += 0 if ok else 1
    print(json.dumps(dict(case=case, variant=variant, direct=direct, T=T, training=training, clipped=clipped, n_max=n_max,
                          err=err, flags_ok=bool(ok_flags), ok=bool(ok))), flush=True)
print(json.dumps(dict(cases=n_cases, mismatching=bad, worst_err=float(worst))))
sys.exit(1 if bad else 0)
