#!/usr/bin/env python3
"""bench.py -- quadrotor env-steps/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over this rank's batch: one `quad.step()` for every env.
Default workload = BASELINE.json configs[1]: 65 536 parallel envs per GPU, zero action from a
(perturbed) hover, dynamics only, the reference's integrator (scipy-exact adaptive RK45).  Envs
shard across ranks with no data-path collective (weak scaling: per-GPU batch fixed); the episode
returns are all-gathered once after the timed region for the configs that have rewards.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per env-step, fp32 SoA (SURVEY.md section 8d / DESIGN.md)
BYTES_DYNAMICS = 104   # read state 13x4 + write state 13x4
BYTES_FULL = 224       # full quad.step (reads 88 + writes 136)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (envs/GPU, direct_control, controller, flags, bytes/env-step, description)
    "config2": (65536, 1, "const", 1, BYTES_DYNAMICS,
                "BASELINE configs[1]: 65536 envs/GPU, zero action, perturbed hover, dynamics only"),
    "config3": (262144, 0, "lqr", 2, BYTES_FULL,
                "BASELINE configs[2]: 262144 envs/GPU, LQR inner loop, random init, full step"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    p.add_argument("--envs", type=int, default=0, help="envs per GPU (0 = the workload's own size)")
    p.add_argument("--integrator", default="rk45", choices=["rk45", "dp5", "rk4"])
    p.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    p.add_argument("--launch", default="fused", choices=["step", "fused"],
                   help="step: one kernel launch per env step (quad.step API); fused: the whole K-step "
                        "region in ONE launch with the state held in registers (qenv_rollout)")
    p.add_argument("--chunk", type=int, default=0, help="env steps per graph replay / fused launch "
                   "(0 = gcd(steps, warmup) capped at 250)")
    p.add_argument("--no-graph", action="store_true", help="launch=step: eager launches instead of hipGraph replay")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--extras", action="store_true", help="also time the other integrators / launch modes / sizes")
    return p.parse_args()


def make_init(workload, n_global, lo, hi):
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    x0 = fast_random_init_states(2 if workload == "config2" else 3, n_global, lo, hi)
    if workload == "config2":  # hover + N(0, 0.1) velocities and body rates
        x0[:, 0:5:2] = 0.0
        x0[:, 1:6:2] *= 0.05
        x0[:, 6:10] = [1, 0, 0, 0]
        x0[:, 10:13] *= 0.05
    return x0


def make_runner(env, controller, flags, launch, chunk, use_graph):
    """Returns (run_chunk, steps_per_call).  launch='step': `chunk` quad.step launches per call,
    captured once into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the
    stream the env launches on) so the launch-bound inner loop replays without host work."""
    if launch == "fused":
        return (lambda: env.rollout(controller, chunk, flags=flags)), chunk
    if controller == "const":
        def body():
            env.step(None, flags=flags)
    else:
        act = torch.empty(4, env.n_envs, device=env.device, dtype=env.dtype)
        out = (torch.empty(14, env.n_envs, device=env.device, dtype=env.dtype),
               torch.empty(env.n_envs, device=env.device, dtype=env.dtype),
               torch.empty(env.n_envs, device=env.device, dtype=torch.uint8))

        def body():
            env.lqr_action(act) if controller == "lqr" else env.pid_action(None, act)
            env.step(act, flags=flags, out=out)
    if not use_graph:
        def run():
            for _ in range(chunk):
                body()
        return run, chunk
    side = torch.cuda.Stream(device=env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream(env.device).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for _ in range(chunk):
            body()
    return graph.replay, chunk


def time_region(run, per_call, steps, world, dev):
    """barrier + sync | exactly `steps` env steps | barrier + sync; returns seconds (max over ranks)."""
    assert steps % per_call == 0
    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps // per_call):
        run()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_launch_us(run, per_call, launch, reps=20):
    """Average duration of ONE launch of the dominant kernel from HIP events on the launch stream
    (the env launches on torch's current stream, so torch.cuda.Event brackets it).  launch='fused':
    one event pair per launch.  launch='step': the kernels are a few microseconds long, less than
    an event pair costs, so the pair brackets one hipGraph replay of `per_call` back-to-back
    launches and the result is divided by per_call (it therefore includes the ~0.5 us dependent
    launch boundary, which makes `achieved` conservative)."""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for a, b in evs:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    us = float(np.mean(ts[: max(1, len(ts) * 3 // 4)]))  # drop the slowest quarter (clock ramps, interrupts)
    return us if launch == "fused" else us / per_call


def cpu_baseline(workload, integrator, seconds):
    """The fp64 C oracle (a port of the reference's algorithm, pinned by the reference's golden
    vectors) timed on this host's cores on a bounded sample of the same workload."""
    from oracle import quad_oracle as qo
    qo.build()
    _, direct, controller, _, _, _ = WORKLOADS[workload]
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the GPU box's CPU share for one GPU is 16
    ctrl = {"const": 0, "lqr": 1, "pid": 2}[controller]
    x0 = make_init(workload, 4096 * threads, 0, 4096 * threads)
    # calibrate, then size the sample for ~`seconds` of wall time
    t0 = time.perf_counter()
    qo.batch_rollout(x0[:64], 50, direct=direct, integrator=integrator, controller=ctrl)
    rate1 = 64 * 50 / (time.perf_counter() - t0)
    steps = 1000
    per_thread = int(max(64, min(4096, rate1 * seconds / steps)))
    chunks = [x0[k * per_thread:(k + 1) * per_thread] for k in range(threads)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL inside the C call
        list(ex.map(lambda c: qo.batch_rollout(c, steps, direct=direct, integrator=integrator, controller=ctrl), chunks))
    dt = time.perf_counter() - t0
    total = threads * per_thread * steps
    return {"value": total / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads x {per_thread} envs x {steps} steps of {workload} ({integrator}, fp64 C oracle), "
                      f"{dt:.1f} s; single-thread rate {rate1:.0f}/s; reference Python measured in the survey "
                      f"container: ~800 env-steps/s/core (BASELINE.md section 3)"}


def main():
    args = parse()
    from autonomous_quadrotor_environment_amd import BatchedQuad
    from autonomous_quadrotor_environment_amd.distributed import (all_gather_returns, init_process_group_from_env,
                                                                  shard_range)
    rank, world, local = init_process_group_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    envs_per_gpu, direct, controller, flags, bytes_per_step, desc = WORKLOADS[args.workload]
    if args.envs:
        envs_per_gpu = args.envs
    n_global = envs_per_gpu * world
    lo, hi = shard_range(n_global, rank, world)
    dt = torch.float32 if args.dtype == "f32" else torch.float64

    def build_env(integrator):
        env = BatchedQuad(hi - lo, 0.01, 10 ** 9, training=False, direct_control=direct, T=1,
                          integrator=integrator, device=dev, dtype=dt)
        env.reset(torch.from_numpy(np.ascontiguousarray(make_init(args.workload, n_global, lo, hi).T)))
        if controller == "pid":
            env.pid_reset()
        return env

    chunk = args.chunk or max(c for c in range(1, 251) if args.steps % c == 0 and (args.warmup % c == 0))
    if args.steps % chunk or args.warmup % chunk:
        raise SystemExit("--chunk must divide --steps and --warmup")
    env = build_env(args.integrator)
    run, per_call = make_runner(env, controller, flags, args.launch, chunk, not args.no_graph)
    time_region(run, per_call, args.warmup, world, dev)         # W untimed warm-up steps
    secs = time_region(run, per_call, args.steps, world, dev)   # exactly K timed steps
    value = n_global * args.steps / secs
    finite = bool(torch.isfinite(env.state).all())
    if flags != 1:
        all_gather_returns(env.episode_return, n_global)  # the path's one exchange step (RCCL over xGMI)

    # dominant kernel, per-launch duration by HIP events on the launch stream
    spl = chunk if args.launch == "fused" else 1
    k_us = kernel_launch_us(run, per_call, args.launch)
    esz = 1 if args.dtype == "f32" else 2
    algo_bytes = bytes_per_step * esz * (hi - lo) * spl
    achieved = algo_bytes / (k_us * 1e-6) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(f"{args.workload}:{args.integrator}:{args.launch}:{args.dtype}")
        except Exception:
            traffic = None
    kname = ("k_rollout" if (args.launch == "fused" or controller != "const") else "k_step") + \
            f"<{'float' if args.dtype == 'f32' else 'double'},{args.integrator}>"

    out = {
        "metric": "quadrotor env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": secs / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.dtype == "f32" else "f64", "data": "synthetic",
        "config": {"workload": desc, "envs_per_gpu": hi - lo, "global_envs": n_global,
                   "integrator": args.integrator, "launch": args.launch, "controller": controller,
                   "steps_per_graph_replay_or_launch": chunk, "hip_graph": (args.launch == "step" and not args.no_graph),
                   "parallelism": f"env-sharded x{world}", "finite": finite},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": kname,
                     "kernel_us": k_us, "algorithmic_bytes_per_launch": algo_bytes,
                     "bytes_per_env_step": bytes_per_step * esz, "env_steps_per_launch": (hi - lo) * spl},
    }

    if args.extras and world == 1:
        extras = {}
        for integ in ("rk45", "dp5", "rk4"):
            e2 = build_env(integ)
            for launch in ("step", "fused"):
                r2, pc = make_runner(e2, controller, flags, launch, chunk, not args.no_graph)
                time_region(r2, pc, chunk, 1, dev)
                s = time_region(r2, pc, args.steps, 1, dev)
                extras[f"{integ}:{launch}"] = n_global * args.steps / s
            del e2
        out["extras_env_steps_per_s"] = extras

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.integrator, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
