"""GPU box: wall time of one masked reset (k_reset + the T warm-up steps on the compacted list) against the number of
envs being reset -- the serial tail of every PPO rollout step (config 5)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
lean = len(sys.argv) > 2 and sys.argv[2] == "lean"
env = aq.BatchedQuad(N, 0.01, 1000, training=True, direct_control=1, T=5, dtype=torch.float64 if lean else torch.float32,
                     compute="lean" if lean else None)
env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
for m in (0, 1, 64, 300, 1024, 8192, N):
    mask = torch.zeros(N, dtype=torch.uint8, device="cuda")
    if m:
        mask[torch.randperm(N, device="cuda")[:m]] = 1
    for _ in range(20):
        env.reset_sampled(1, 1, 0, mask=mask, push_window=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(200):
        env.reset_sampled(1, k + 2, 0, mask=mask, push_window=True)
    b.record(); torch.cuda.synchronize()
    print(json.dumps(dict(n_envs=N, handle="lean" if lean else "f32", warmup_lanes=os.environ.get("QENV_WARMUP_LANES", "8"), n_reset=m, us_per_masked_reset=a.elapsed_time(b) * 1000 / 200)), flush=True)
