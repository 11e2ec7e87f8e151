"""GPU box: cost of one fused env step (constant action) by step mode -- what the reward / Euler / bookkeeping tail of
a full step adds to the RK45 itself, at 1 and 4 waves per SIMD."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

for N in (65536, 262144):
    for direct in (1, 0):
        for name, flags in (("dynamics_only", L.STEP_DYNAMICS_ONLY), ("no_aux", L.STEP_NO_AUX), ("full", L.STEP_FULL)):
            env = aq.BatchedQuad(N, 0.01, 10 ** 9, training=False, direct_control=direct, T=5)
            env.reset(env.sample_init_states(0, 0, 0), return_history=False)
            act = [0.0] * 4 if direct else [1.03 * 9.82, 0, 0, 0]
            env.rollout("const", 200, action=act, flags=flags)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            env.rollout("const", 2000, action=act, flags=flags)
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1000 / 2000
            print(json.dumps(dict(n_envs=N, direct=direct, mode=name, us_per_step=round(us, 3), G_env_steps_s=round(N / us / 1e3, 2))), flush=True)
