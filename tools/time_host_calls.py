"""GPU box: host-side cost per C-ABI call (enqueue only, no synchronisation inside the loop) on a tiny batch where the GPU
keeps up: what a rollout step costs the host thread when several stream shards are stepped from one thread."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = aq.ActorCriticWeights.random(128, 75, 4, device="cuda", seed=1); w.std = 0.1
env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5, dtype=torch.float64, compute="lean")
ro = aq.PolicyRollout(env, w, 64, store_obs=True, auto_reset=True, seed=1)
ro.start()
act = torch.zeros(4, N, device="cuda", dtype=torch.float64)
out = (None, torch.empty(N, device="cuda", dtype=torch.float64), torch.empty(N, device="cuda", dtype=torch.uint8))
mean = torch.empty(4, N, device="cuda", dtype=torch.float64)


def timeit(name, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(json.dumps(dict(call=name, envs=N, host_us=round((t1 - t0) / n * 1e6, 2), total_us=round((t2 - t0) / n * 1e6, 2))), flush=True)


timeit("mlp_forward", lambda: env.mlp_forward(w, "actor", out=mean))
timeit("step(no_aux, push)", lambda: env.step(act, flags=2, out=out, push_window=True))
timeit("rollout(const, 1)", lambda: env.rollout("const", 1))


def ppo():
    ro.t = 0
    ro.step()
timeit("PolicyRollout.step (ppo_step)", ppo)
