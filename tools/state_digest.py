"""GPU box: SHA-256 of the fp32 workspace state after fixed rollouts -- run before and after a kernel change that is
meant to be bit-neutral (packing, scheduling, register allocation) and compare the lines."""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

def dig(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]

out = {}
for integ in ("rk45", "dp5", "rk4"):
    for ctrl, direct in (("const", 1), ("lqr", 0), ("pid", 0)):
        for dt in (torch.float32, torch.float64):
            env = aq.BatchedQuad(4096, 0.01, 10 ** 9, training=False, direct_control=direct, T=5, integrator=integ, dtype=dt)
            env.reset(env.sample_init_states(3, 0, 0), return_history=False)
            env.rollout(ctrl, 300, flags=L.STEP_FULL)
            for _ in range(3):
                env.step(None)
            out[f"{integ}:{ctrl}:{'f32' if dt == torch.float32 else 'f64'}"] = dig(env.state) + dig(env.episode_return)
env = aq.BatchedQuad(4096, 0.01, 10 ** 9, training=False, direct_control=1, T=1, dtype=torch.float64, compute="f32")
env.reset(env.sample_init_states(3, 0, 0), return_history=False)
env.rollout("const", 300, flags=L.STEP_DYNAMICS_ONLY)
out["rk45:const:mixed"] = dig(env.state)
print(json.dumps(out, sort_keys=True))
