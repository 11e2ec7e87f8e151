"""GPU box: config 5's rollout step -- host enqueue time per step (no synchronisation inside the loop) against the
GPU time per step: tells whether the Python / ctypes launch path or the kernels bound the PPO rollout."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from autonomous_quadrotor_environment_amd import ActorCriticWeights, PolicyRollout

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = bench.build_env("config5", "lean", "rk45", n, 0, n, dev)
w = ActorCriticWeights.random(128, 75, 4, device=dev, seed=1); w.std = 0.1
ro = PolicyRollout(env, w, 128, store_obs=True, auto_reset=True, seed=4, env_offset=0)
for rep in range(3):
    ro.t = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(128):
        ro.step()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps(dict(rep=rep, host_us_per_step=(t1 - t0) / 128 * 1e6, total_us_per_step=(t2 - t0) / 128 * 1e6)))
