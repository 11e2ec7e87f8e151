"""GPU box: a bare driver of 250-step fused rollouts for counter collection (tools/pmc_valu_count.sh with PMC_CMD):
python tools/rollout_driver.py CONTROLLER FLAGS INTEGRATOR [envs] [handle]   (CONTROLLER const|lqr|pid, FLAGS 0 full|1 dynamics|2 no-aux)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states

ctrl, flags, integ = sys.argv[1], int(sys.argv[2]), sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
handle = sys.argv[5] if len(sys.argv) > 5 else "lean"
kw = dict(dtype=torch.float64, compute="lean") if handle == "lean" else (dict(dtype=torch.float64) if handle == "f64" else dict(dtype=torch.float32))
env = aq.BatchedQuad(n, 0.01, 10 ** 9, training=False, direct_control=1 if ctrl == "const" else 0, T=1, integrator=integ, **kw)
env.reset(torch.from_numpy(np.ascontiguousarray(fast_random_init_states(3, n).T)))
if ctrl == "pid":
    env.pid_reset()
for _ in range(8):
    env.rollout(ctrl, 250, flags=flags)
torch.cuda.synchronize()
