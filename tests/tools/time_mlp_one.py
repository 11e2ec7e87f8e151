"""GPU box: launch time of the actor forward (75 -> 128 -> 128 -> 4, T = 5) at one batch size; QENV_LIB selects an A/B
build of the library (developer ablations).  python tests/tools/time_mlp_one.py [envs]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import autonomous_quadrotor_environment_amd as aq

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
w = aq.ActorCriticWeights.random(128, 75, 4, device="cuda", seed=1)
env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5)
env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
out = torch.empty(4, N, device="cuda", dtype=env.dtype)
for _ in range(50):
    env.mlp_forward(w, "actor", out=out)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        env.mlp_forward(w, "actor", out=out)
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) * 1000 / 200)
print(json.dumps(dict(lib=os.environ.get("QENV_LIB", "default"), envs=N, us=round(best, 2))))
