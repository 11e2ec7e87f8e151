"""GPU box: the learner's kernels at training-demo size (256 envs x 32 steps = 8 192 samples): latency, not throughput."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

N, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256), 32
w = aq.ActorCriticWeights.random(128, 75, 4, seed=1)
obs = torch.randn(n, 75, N, device="cuda")
act = torch.randn(n, 4, N, device="cuda") * 0.1
lp = torch.randn(n, 4, N, device="cuda")
adv = torch.randn(n, N, device="cuda"); ret = torch.randn(n, N, device="cuda")
ln = aq.PpoLearner(w, K_epochs=1)
for fused in (True, False):
    ln.fused_loss = fused
    for _ in range(5):
        ln.gradients(obs, act, lp, adv, ret)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ln.gradients(obs, act, lp, adv, ret)
    b.record(); torch.cuda.synchronize()
    print(json.dumps(dict(n_envs=N, steps=n, fused_loss=fused, us_per_gradients_call=round(a.elapsed_time(b) * 1000 / 50, 1))), flush=True)

# a whole update (K epochs of gradients + Adam) eager vs replayed from a hipGraph
import types, time
ro = types.SimpleNamespace(t=n, obs=obs, actions=act, logprobs=lp)
for graph in (False, True):
    ln2 = aq.PpoLearner(aq.ActorCriticWeights.random(128, 75, 4, seed=1), K_epochs=10, group=False)
    for _ in range(2):
        ln2.update(ro, ret, adv, graph=graph)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ln2.update(ro, ret, adv, graph=graph)
    torch.cuda.synchronize()
    print(json.dumps(dict(n_envs=N, steps=n, hip_graph=graph, us_per_epoch=round((time.perf_counter() - t0) / 100 * 1e6, 1))), flush=True)
