"""GPU box: wall time of one PPO update epoch (forward, loss, backward, weight gradients, Adam) on a stored rollout
of config-5 size, per phase."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
w = aq.ActorCriticWeights.random(128, 75, 4, seed=1); w.std = 0.1
env = aq.BatchedQuad(N, 0.01, 1000, training=True, direct_control=1, T=5, dtype=torch.float32)
ro = aq.PolicyRollout(env, w, S, store_obs=True, auto_reset=True, seed=3)
ro.start()
t0 = time.time()
for _ in range(S):
    ro.step()
torch.cuda.synchronize(); t_roll = time.time() - t0
ret, adv = ro.finish(ddof=0, group=False, bootstrap=True)
learner = aq.PpoLearner(w, K_epochs=1)
learner.update(ro, ret, adv)  # warm-up (buffers)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(3):
    learner.update(ro, ret, adv)
torch.cuda.synchronize()
t_up = (time.time() - t0) / 3
samples = N * S
flops = samples * 2 * (75 * 128 + 128 * 128 + 128 * 4 + 75 * 128 + 128 * 128 + 128) * 3  # fwd + 2x for bwd, both heads
print(json.dumps(dict(envs=N, horizon=S, samples=samples, rollout_s=round(t_roll, 3), rollout_Msteps_s=round(samples / t_roll / 1e6, 1),
                      update_epoch_s=round(t_up, 4), update_Msamples_s=round(samples / t_up / 1e6, 1),
                      update_TFLOPs=round(flops / t_up / 1e12, 1))))
