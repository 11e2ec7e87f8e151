"""GPU box: config 2 (perturbed hover, zero action, dynamics only) -- deviation of every (integrator, precision)
from the scipy-exact fp64 handle (= the reference to 1e-11) after 100 / 500 / 1000 steps, scaled metric."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import autonomous_quadrotor_environment_amd as aq

N = 4096
x0 = np.ascontiguousarray(bench.make_init("config2", N, 0, N).T)
scale = torch.tensor([5, 10, 5, 10, 5, 10, 1, 1, 1, 1, 20, 20, 20.0], device="cuda", dtype=torch.float64).reshape(13, 1)

def run(integ, dt, compute=None):
    env = aq.BatchedQuad(N, 0.01, 10**6, direct_control=1, T=1, integrator=integ, dtype=dt, compute=compute)
    env.reset(torch.from_numpy(x0))
    out = {}
    done = 0
    for n in (100, 500, 1000):
        env.rollout("const", n - done, flags=1)
        done = n
        out[n] = env.state.double().clone()
    return out
ref = run("rk45", torch.float64)
for integ in ("rk45", "dp5", "rk4"):
    for name, dt, comp in (("f32", torch.float32, None), ("mix", torch.float64, "f32"), ("f64", torch.float64, None)):
        got = run(integ, dt, comp)
        err = {n: float(((got[n] - ref[n]).abs() / torch.maximum(ref[n].abs(), 1e-3 * scale)).max()) for n in got}
        err_abs = {n: float(((got[n] - ref[n]).abs() / scale).max()) for n in got}  # absolute error in bounding-box units
        worst = {n: int(((got[n] - ref[n]).abs() / torch.maximum(ref[n].abs(), 1e-3 * scale)).max(1).values.argmax()) for n in got}
        print(json.dumps(dict(integrator=integ, handle=name, scaled_err=err, abs_over_scale=err_abs, worst_component=worst)), flush=True)
