"""GPU box: where the env step inside k_ppo_fused spends its cycles (diagnostic build:
tools/build_variant.sh estamps "-DQENV_FUSED_STAMPS -DQENV_ENVSTEP_STAMPS" qenv_env_lean.hip).

    QENV_LIB=build_variants/libqenv_estamps.so python tests/tools/envstep_stamps.py [envs] [sample 0/1]
"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "g9_mlp_N128.npz"))
w = aq.ActorCriticWeights.from_state_dict({k: g[k] for k in g.files if k.startswith(("actor", "critic"))})
w.std = 0.1
env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5, dtype=torch.float64, compute="lean")
ro = aq.PolicyRollout(env, w, 32, store_obs=True, auto_reset=True, seed=5, inband=True) if sample else aq.PolicyRollout(env, w.actor_only(), 32)
ro.start()
for _ in range(3):
    ro.t = 0
    ro.run(32) if sample else ro.run_deterministic(32)
torch.cuda.synchronize()
fn = L.lib.qenv_debug_fused_stamps
fn.restype = C.c_int
buf = (C.c_ulonglong * (1024 * 16))()
assert fn(buf, 1024 * 16) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
s = s[s[:, 0] > 0]
pairs = [("prepare_action", 7, 10), ("integrate", 10, 11), ("euler_update", 11, 12), ("done_reward", 12, 13), ("stores + restart", 13, 8),
         ("env step", 7, 8), ("window shift", 8, 9), ("whole step", 0, 9)]
rows = {nm: [float(np.percentile(s[:, b] - s[:, a], q)) for q in (10, 50, 90)] for nm, a, b in pairs}
print(json.dumps(dict(envs=N, sample=sample, waves=len(s), phases_p10_p50_p90=rows), indent=1))
