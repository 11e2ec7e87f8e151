"""GPU box: config 4 (deterministic actor in the loop, no resets) as S env shards on S streams of one GPU, eager launches:
wall time per global step for S = 1, 2, 4 (the single batch through a hipGraph replay is bench.py's own figure)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from autonomous_quadrotor_environment_amd import ActorCriticWeights, StreamShardedRollout

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
H = 100
w = ActorCriticWeights.random(128, 75, 4, device=dev, seed=1)
w = ActorCriticWeights(w.actor, None, dev)
for S in (1, 2, 4):
    make_env = lambda lo, hi: bench.build_env("config4", "lean", "rk45", n, lo, hi, dev)
    ro = StreamShardedRollout(make_env, n, w, H, shards=S, device=dev)
    ts = []
    for rep in range(4):
        ro.t = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(H):
            ro.step(deterministic=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) / H * 1e6, (t2 - t0) / H * 1e6))
    print(json.dumps(dict(envs=n, shards=S, host_us_per_step=round(ts[-1][0], 1), us_per_step=round(min(t[1] for t in ts), 1),
                          G_env_steps_per_s=round(n / min(t[1] for t in ts) / 1e3, 3))), flush=True)
    del ro
