"""GPU box: wall time of one PPO rollout step at training-demo size (256 envs) -- host launch path vs kernel time
(run under rocprofv3 --kernel-trace --stats for the kernel side)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = 32
w = aq.ActorCriticWeights.random(128, 75, 4, device="cuda", seed=1)
env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5, dtype=torch.float32)
ro = aq.PolicyRollout(env, w, H, store_obs=True, auto_reset=True, seed=4)
ro.start()


def horizon():
    ro.t = 0
    for _ in range(H):
        ro.step()
    ro.finish()
for _ in range(5):
    horizon()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    horizon()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(json.dumps(dict(n_envs=N, us_per_step_wall=round((t2 - t0) / (50 * H) * 1e6, 1),
                      us_per_step_host_enqueue=round((t1 - t0) / (50 * H) * 1e6, 1))))
