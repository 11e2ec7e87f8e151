"""GPU box: config 5's rollout as S independent env shards on S streams of ONE GPU (each shard its own handle; the env
trajectories are shard-invariant, so the union of the shards is bit-identical to the single-batch run): while one shard
sits in the latency-bound warm-up of its finished envs (a handful of wavefronts, ~60 us), the others' policy forward and
env step fill the chip.  Prints the wall time per global step for S = 1, 2, 4, 8."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from autonomous_quadrotor_environment_amd import ActorCriticWeights, PolicyRollout

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
H = 64
for overlap_env in ("0",):
    os.environ["QENV_PPO_OVERLAP"] = overlap_env
    for S in (1, 2, 4, 8):
        w = ActorCriticWeights.random(128, 75, 4, device=dev, seed=1); w.std = 0.1
        per = n // S
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        ros = []
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                env = bench.build_env("config5", "lean", "rk45", n, k * per, (k + 1) * per, dev)
                ros.append(PolicyRollout(env, w, H, store_obs=True, auto_reset=True, seed=4, env_offset=k * per))
        for fused in (False, True):
            for ro in ros:
                ro.overlap = fused
            ts = []
            for rep in range(3):
                for ro in ros:
                    ro.t = 0
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(H):
                    for k in range(S):
                        with torch.cuda.stream(streams[k]):
                            ros[k].step()
                t1 = time.perf_counter()
                for k in range(S):
                    with torch.cuda.stream(streams[k]):
                        ros[k].env.ppo_flush()
                torch.cuda.synchronize(); t2 = time.perf_counter()
                ts.append(((t1 - t0) / H * 1e6, (t2 - t0) / H * 1e6))
            print(json.dumps(dict(shards=S, envs_per_shard=per, fused_call=fused, host_us_per_step=ts[-1][0], us_per_step=min(t[1] for t in ts))), flush=True)
        del ros
