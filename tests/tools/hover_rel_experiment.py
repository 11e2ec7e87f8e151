"""GPU box: the hover-relative fp32 right-hand side (round-4 verdict, item 5; -DQENV_HOVER_REL builds of qenv_env.hip /
qenv_env_mix.hip: tools/build_variant.sh hoverrel "-DQENV_HOVER_REL" qenv_env.hip qenv_env_mix.hip) against the product's fp32
arithmetic, on the two qualifications the verdict names:
  (a) config 2's own distribution, zero action, 100 / 500 / 1 000 open-loop steps (tests/tools/config2_parity.py's metric);
  (b) the PPO closed loop teacher-forced (G6-PPO: the reference's recorded actions through the env, 500 steps, four seeds)
      against fp64 oracle twins.
Run once per library: QENV_LIB=... python tests/tools/hover_rel_experiment.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
import bench
from oracle import quad_oracle as O

O.build()
SC = np.array([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0])
H = {"f32": (torch.float32, None), "mix": (torch.float64, "f32")}
out = {"lib": os.environ.get("QENV_LIB", "product")}
n, sub = 4096, 256
x0 = bench.make_init("config2", n, 0, n)
refs = {u: O.batch_rollout(x0[:sub], u, direct=1, controller=0)[0] for u in (100, 500, 1000)}
for handle in ("mix", "f32"):
    dt, comp = H[handle]
    env = aq.BatchedQuad(n, 0.01, 10 ** 9, training=False, direct_control=1, T=1, dtype=dt, compute=comp)
    env.reset(torch.from_numpy(np.ascontiguousarray(x0.T)))
    done, res = 0, {}
    for upto in (100, 500, 1000):
        while done < upto:
            k = min(250, upto - done)
            env.rollout("const", k, flags=1)
            done += k
        got = env.state[:, :sub].double().cpu().numpy().T
        ev = np.abs(got - refs[upto]) / np.maximum(np.abs(refs[upto]), 1e-3 * SC)
        e, c = np.unravel_index(ev.argmax(), ev.shape)
        res[upto] = dict(scaled=float(ev.max()), p99=float(np.percentile(ev.max(1), 99)), component=int(c))
    out[f"config2_{handle}"] = res
g = [np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", f"g6_ppo128_seed{s}.npz")) for s in (1, 2, 3, 4)]
x0 = np.stack([q["x0"] for q in g], 1)
for handle in ("mix", "f32"):
    dt, comp = H[handle]
    env = aq.BatchedQuad(4, 0.01, 500, training=False, direct_control=1, T=5, dtype=dt, compute=comp)
    env.reset(torch.from_numpy(x0), return_history=False, push_window=True)
    twins = []
    for q in g:
        r = O.OracleQuad(0.01, 500, training=False, direct_control=1, T=5); r.reset(q["x0"]); twins.append(r)
    worst = {}
    for k in range(500):
        rec = np.stack([q["actions"][k] for q in g], 1)
        env.step(torch.from_numpy(rec), flags=2, push_window=True)
        for e, r in enumerate(twins):
            r.step(g[e]["actions"][k])
        if k + 1 in (10, 100, 500):
            st = env.state.double().cpu().numpy().T
            ref = np.array([r.state for r in twins])
            worst[k + 1] = float(np.max(np.abs(st - ref) / np.maximum(np.abs(ref), 1e-3 * SC)))
    out[f"g6_ppo_teacher_forced_{handle}"] = worst
print(json.dumps(out))
