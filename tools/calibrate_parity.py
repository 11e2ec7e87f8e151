"""Developer tool (GPU box): measured parity errors of the HIP path vs the reference's golden
vectors for every case/dtype/integrator -- the numbers the test tolerances and DESIGN.md quote.
Metrics: rel = max |d|/max(|ref|,1e-3); sc = max |d|/max(|ref|, 1e-3*scale_k)
with scale = BB box (pos 5, vel 10, q 1, omega 20)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from tests.conftest import golden

SCALE = np.array([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0])
DT = {"f32": torch.float32, "f64": torch.float64, "mix": torch.float64, "lean": torch.float64}
COMPUTE = {"f32": None, "f64": None, "mix": "f32", "lean": "lean"}
DTYPES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "mix", "lean", "f64"]
INTEGS = sys.argv[2].split(",") if len(sys.argv) > 2 else ["rk45", "dp5", "rk4"]
out = {}

def errs(got, ref):
    d = np.abs(got - ref)
    return float(np.max(d / np.maximum(np.abs(ref), 1e-3))), float(np.max(d / np.maximum(np.abs(ref), 1e-3 * SCALE)))

def dev(x, dt):
    return torch.from_numpy(np.ascontiguousarray(x)).to("cuda", DT[dt])

for dt in DTYPES:
    for integ in INTEGS:
        # one step
        for name, kw in (("g4_step_direct", dict(direct_control=1)), ("g4_step_indirect", dict(direct_control=0))):
            g = golden(name); n = len(g["x0"])
            env = aq.BatchedQuad(n, 0.01, 1000, T=1, dtype=DT[dt], compute=COMPUTE[dt], integrator=integ, **kw)
            env.reset(dev(g["x0"].T, dt)); env.step(dev(g["action"].T, dt))
            st = env.state.double().cpu().numpy().T
            single = (g["nfev_reset"] == 8) & (g["nfev_step"] == 8)
            out[f"onestep:{name}:{dt}:{integ}"] = dict(all=errs(st, g["s_state"]), single_substep=errs(st[single], g["s_state"][single]),
                vq_abs=float(np.max(np.abs(env.V_q.double().cpu().numpy().T - g["s_quat_state"][:, 10:]))),
                ang_abs=float(np.max(np.abs(env.ang.double().cpu().numpy().T - g["s_ang"]))),
                reward_abs=float(np.max(np.abs(env.reward.double().cpu().numpy() - g["s_reward"]))))
        # open loop hover
        for name in ("g6_hover", "g6_hover_perturbed"):
            g = golden(name)
            env = aq.BatchedQuad(64, 0.01, 2000, training=False, direct_control=1, T=1, dtype=DT[dt], compute=COMPUTE[dt], integrator=integ)
            env.reset(dev(np.repeat(g["x0"][:, None], 64, 1), dt))
            r = {}
            for k in range(1000):
                env.step(None, flags=1)
                if k + 1 in (100, 500, 1000):
                    r[k + 1] = errs(env.state.double().cpu().numpy()[:, 0], g["traj"][k, :13])
            out[f"openloop:{name}:{dt}:{integ}"] = r
        # closed loop LQR
        gs = [golden(f"g6_lqr_seed{s}") for s in (1, 2, 3, 4)]
        env = aq.BatchedQuad(4, 0.01, 500, training=True, direct_control=0, T=1, dtype=DT[dt], compute=COMPUTE[dt], integrator=integ)
        env.reset(dev(np.stack([g["x0"] for g in gs], 1), dt))
        rows = []
        for k in range(500):
            env.rollout("lqr", 1)
            rows.append(env.state.double().cpu().numpy().copy())
        rows = np.array(rows)
        out[f"closedloop:lqr:{dt}:{integ}"] = [errs(rows[:, :, e], gs[e]["traj"][:, :13]) for e in range(4)]
        out[f"closedloop:lqr_first100:{dt}:{integ}"] = [errs(rows[:100, :, e], gs[e]["traj"][:100, :13]) for e in range(4)]
        # PID
        g = golden("g6_pid_seed1")
        env = aq.BatchedQuad(1, 0.01, 1000, training=True, direct_control=0, T=5, dtype=DT[dt], compute=COMPUTE[dt], integrator=integ)
        env.reset(dev(g["x0"].reshape(13, 1), dt)); env.pid_reset()
        rows = []
        for k in range(1000):
            env.rollout("pid", 1, pid_first_action=(k == 0))
            rows.append(env.state.double().cpu().numpy()[:, 0].copy())
        out[f"closedloop:pid:{dt}:{integ}"] = errs(np.array(rows), g["traj"][:, :13])
for k, v in out.items():
    print(k, json.dumps(v))
