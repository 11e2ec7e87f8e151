"""GPU box: a bare driver of the fused rollout launches for counter collection (tools/pmc_kernel.sh fused k_ppo_fused
tests/tools/fused_driver.py): config 4's deterministic 50-step launches at 131 072 envs and config 5's sampling 128-step launches
at 65 536 envs, a few of each."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import autonomous_quadrotor_environment_amd as aq

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "g9_mlp_N128.npz"))
w = aq.ActorCriticWeights.from_state_dict({k: g[k] for k in g.files if k.startswith(("actor", "critic"))})
w.std = 0.1
lean = dict(dtype=torch.float64, compute="lean")
env = aq.BatchedQuad(131072, 0.01, 500, training=True, direct_control=1, T=5, **lean)
ro = aq.PolicyRollout(env, aq.ActorCriticWeights(w.actor, None, env.device), 50)
ro.start()
for _ in range(4):
    ro.t = 0
    ro.run_deterministic(50)
env5 = aq.BatchedQuad(65536, 0.01, 500, training=True, direct_control=1, T=5, **lean)
ro5 = aq.PolicyRollout(env5, w, 128, store_obs=True, auto_reset=True, seed=5, inband=True)
ro5.start()
for _ in range(3):
    ro5.t = 0
    ro5.run(128)
torch.cuda.synchronize()
print("done")
