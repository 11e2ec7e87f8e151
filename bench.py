#!/usr/bin/env python3
"""bench.py -- quadrotor env-steps/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over this rank's batch: one `quad.step()` for every env.
Default workload = BASELINE.json configs[2], the largest single-GPU configuration: 262 144 parallel
envs per GPU, the LQR inner loop (lqr_quad.py:123-168) on random initial states (seed 3), the full
step (integrator, Euler angles, done, reward), the reference's integrator (scipy-exact adaptive RK45),
500-step episodes, the envs placed so that the ones the LQR never stabilises share wavefronts (BatchedQuad.regroup after
one untimed placement episode: per-env results bit-identical; `configs.config3_natural_order` = the batch as drawn;
--placement natural).  (Rounds 1-4 quoted configs[1], the smaller one; it is now an entry of `configs`, perturbed and
literally-at-hover.)  Envs shard across ranks with no data-path collective
(weak scaling: per-GPU batch fixed); the episode returns are all-gathered once after the timed region
for the configs that have rewards.

Rank 0 prints TWO lines: first `{"bench_detail": ...}` (every slice in full: configs, next_rows, extras),
then -- last, so that a tail of stdout holds it whole -- the ONE contract line `{"metric": ...}` with the
headline, its roofline, the CPU baseline and a compact copy of every slice.

N > 1: one process per GPU over RCCL.  Under `python -m torch.distributed.run` the ranks come
from the environment; started plainly with `--gpus N` this process only launches those N ranks
itself (before any GPU call), relays rank 0's line and exits with their status -- it never
reports a single-rank number for an N-GPU request.

Every workload runs on the cheapest handle that holds the north-star tolerance (1e-5, scaled
metric of tests/conftest.py) over the workload's own horizon on the workload's own initial
distribution (tests/tools/config2_parity.py): that is the lean fp64 handle
for all of them -- fp32 arithmetic (fp32 and mixed handles) is at 1e-4 / 5e-5 after 100 steps of
config 2 already (thrust minus weight cancels to 1e-6 m/s^2 in fp32).

The K-step region (barrier + sync on both sides, exactly K steps) is repeated until >= 0.3 s
have been timed; `ms_per_step` / `value` are the MEDIAN region -- for the episodic workloads (config 3 / 4) the MEAN over
regions that cover whole episodes: a region costs what its phase of the episode costs.  The contract line carries
the contract fields plus `roofline`, `cpu_baseline`, `sustained` and (default run)
`configs`: short per-GPU slices of the other BASELINE configs.  A rank's clock for a region stops when its own GPU
has finished (torch.cuda.synchronize); the MAX all-reduce over the ranks closes the bracket afterwards.

--workload config5 can collect the rank's env range as `--stream-shards` S shards on their own HIP streams (default 1: since
the rollout is ONE fused launch per horizon the shards gain nothing)
(StreamShardedRollout: bit-identical to one batch; +12 % with round 2's list schedule).  --policy trained|random: the actor (and critic) of configs 4 / 5
are the reference's checked-in N = 128 weights (solved/nn_old_solved_128_*.pth re-exported as arrays) or
nn.Linear-style random-init weights (the reset-heavy variant: that policy drives envs out of the box within tens of steps).  --no-sustained / --no-extras / --no-subconfigs /
--no-cpu-baseline trim the line for profiling runs (tools/profile_round.sh).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per env-step, fp32 SoA (SURVEY.md section 8d / DESIGN.md) -- the same figure for every handle
BYTES_DYNAMICS = 104   # read state 13x4 + write state 13x4
BYTES_FULL = 224       # full quad.step (reads 88 + writes 136)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
MIN_TIMED_S = 0.3
EPISODE_STEPS = 500     # episode length of the closed-loop workloads (lqr_quad.py, ppo.py:301)

WORKLOADS = {
    # name: (envs/GPU, direct_control, controller, flags, bytes/env-step, qualified handle, description)
    "config2": (65536, 1, "const", 1, BYTES_DYNAMICS, "lean",
                "BASELINE configs[1]: 65536 envs/GPU, zero action, perturbed hover, dynamics only"),
    "config2_hover": (65536, 1, "const", 1, BYTES_DYNAMICS, "lean",
                      "BASELINE configs[1] literally: 65536 envs/GPU, zero action from EXACT hover (f0 = 0: scipy accepts ~5 growing "
                      "sub-steps = 32 right-hand sides of a state that does not move; integrate_rk45 recognises the equilibrium and "
                      "returns the same y without them), dynamics only"),
    "config3": (262144, 0, "lqr", 2, BYTES_FULL, "lean",
                "BASELINE configs[2]: 262144 envs/GPU, LQR inner loop, random init (seed 3), full step, 500-step episodes"),
    # PPO workloads: controller names select the runner below; bytes from SURVEY.md section 8d
    "config4": (131072, 1, "ppo_actor", 2, BYTES_FULL + 360, "lean",
                "BASELINE configs[3]: 131072 envs/GPU (1048576 on 8), PPO actor MLP 75-128-128-4 (matrix cores, fp32-accurate) in the "
                "loop, T=5 window, deterministic mean action (ppo_quad_eval.py:44-62), random initial states"),
    "config5": (65536, 1, "ppo_rollout", 2, BYTES_FULL + 360 + 332 + 20, "lean",
                "BASELINE configs[4]: 65536 envs/GPU (524288 on 8), full PPO rollout (actor+critic, Normal(mean,0.1) "
                "Philox sampling, reward_fn, masked auto-reset) + GAE per 128-step horizon, returns all-gathered"),
}
MLP_FLOPS = 2 * (75 * 128 + 128 * 128 + 128 * 4)  # per env and forward: 53.0 kflop
MFMA_F32_PEAK_TFLOPS = 157.3
MAX_CLOCK_GHZ = 2.4
# matrix-core instructions k_mlp<128,5,BF16=2> executes per 32-env tile: name -> (count, pipe cycles each, flops each)
# layer 1: 4 output tiles x 5 k-steps x 6 split products; layer 2: 4 x 4 x 2 x 6 (k_mlp_v3, csrc/qenv_ppo.hip)
MLP_MFMA = {"v_mfma_f32_32x32x16_bf16": (120 + 192, 32, 2 * 32 * 32 * 16)}   # round 3: layer 3 runs on the vector lanes
DEFAULT_WORKLOAD = "config3"   # BASELINE's largest single-GPU configuration (rounds 1-4: config2)
HANDLES = {  # --dtype -> (torch dtype of the rows, compute mode, arithmetic type reported as `dtype`)
    "f32": (torch.float32, None, "f32"),
    "mix": (torch.float64, "f32", "f32"),
    "lean": (torch.float64, "lean", "f64"),
    "f64": (torch.float64, None, "f64"),
}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20000)
    p.add_argument("--warmup", type=int, default=2000)
    p.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    p.add_argument("--envs", type=int, default=0, help="envs per GPU (0 = the workload's own size)")
    p.add_argument("--integrator", default="rk45", choices=["rk45", "dp5", "rk4"])
    p.add_argument("--dtype", default="auto", choices=["auto", "f32", "f64", "mix", "lean"],
                   help="auto = the workload's parity-qualified handle; mix = fp64 rows + fp32 arithmetic "
                        "(QENV_COMPUTE_F32); lean = fp64 multiply-adds, Newton-refined hardware reciprocals, "
                        "guard-banded fast step control (QENV_COMPUTE_LEAN)")
    p.add_argument("--launch", default="fused", choices=["step", "fused"],
                   help="step: one kernel launch per env step (quad.step API); fused: the whole K-step "
                        "region in launches of up to 250 steps with the state held in registers (qenv_rollout)")
    p.add_argument("--chunk", type=int, default=0, help="env steps per graph replay / fused launch "
                   "(0 = the largest divisor of --steps up to 250)")
    p.add_argument("--no-graph", action="store_true", help="launch=step: eager launches instead of hipGraph replay")
    p.add_argument("--stream-shards", type=int, default=1,
                   help="config5: the rank's env range as this many shards on their own HIP streams (1 = one batch)")
    p.add_argument("--policy", default="trained", choices=["trained", "random"],
                   help="configs 4 / 5: the reference's checked-in N=128 weights (default, SURVEY.md 8d) or random-init weights")
    p.add_argument("--schedule", default="inband", choices=["inband", "list"],
                   help="config5: inband = a finished env restarts in the step kernel and its T warm-up steps ride the next T "
                        "regular steps (rows flagged invalid; a step is policy + env step); list = round 2's schedule (a "
                        "warm-up launch over the list of finished envs)")
    p.add_argument("--placement", default="regroup", choices=["regroup", "natural"],
                   help="controller rollouts (config 3): regroup = the envs sorted by |omega| after one untimed placement episode "
                        "(BatchedQuad.regroup); natural = the batch as drawn")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--extras", action="store_true", help="also time the other integrators / launch modes")
    p.add_argument("--no-extras", action="store_true", help="skip the extras that the default 1-GPU config-2 run adds")
    p.add_argument("--no-sustained", action="store_true", help="skip the 250-step-launch `sustained` figure (profiling runs: "
                   "keeps the kernel trace to the K-step launches)")
    p.add_argument("--no-subconfigs", action="store_true", help="skip the config3/4/5 slices of the default 1-GPU run")
    p.add_argument("--min-seconds", type=float, default=MIN_TIMED_S, help="repeat the K-step region until this much is timed")
    return p.parse_args(argv)


# ---- N > 1 without torchrun: launch the ranks, relay rank 0's line ---------------------------------------------------
def visible_gpus():
    """GPUs this process could use, WITHOUT touching the HIP runtime (the launcher parent must stay HIP-free: its children
    initialise the devices): KFD topology nodes that have SIMDs, narrowed by HIP/ROCR_VISIBLE_DEVICES.  None = unknown."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return 0   # no KFD driver: no AMD GPU on this host
    try:
        n = 0
        for d in os.listdir(root):
            with open(os.path.join(root, d, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except Exception:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """Parent of a plain `bench.py --gpus N`: NO HIP / torch.cuda call happens in this process -- the devices are counted
    from sysfs -- and the N ranks start as children through torch.distributed.run."""
    backend = os.environ.get("QENV_DIST_BACKEND") or "nccl"
    ndev = visible_gpus()
    if backend == "nccl" and ndev is not None and ndev < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {ndev} GPU(s) visible (one rank per GPU over RCCL; "
                         f"QENV_DIST_BACKEND=gloo rehearses more ranks than GPUs)")
    with socket.socket() as s:  # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        elif ln.startswith('{"bench_detail"'):
            print(ln.strip(), flush=True)
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0 or line is None:
        raise SystemExit(f"bench.py: the {args.gpus}-rank launch failed (exit {rc}, result line {'present' if line else 'missing'})")
    if json.loads(line)["n_gpus"] != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, the result line reports {json.loads(line)['n_gpus']}")
    print(line, flush=True)
    raise SystemExit(0)


# ---- workload construction ----------------------------------------------------------------------------------------
def make_init(workload, n_global, lo, hi):
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    x0 = fast_random_init_states(2 if workload.startswith("config2") else 3, n_global, lo, hi)
    if workload == "config2":  # hover + N(0, 0.1) velocities and body rates
        x0[:, 0:5:2] = 0.0
        x0[:, 1:6:2] *= 0.05
        x0[:, 6:10] = [1, 0, 0, 0]
        x0[:, 10:13] *= 0.05
    elif workload == "config2_hover":  # det_state = [0,0,0,0,0,0,1,0,0,0,0,0,0] for every env
        x0[:] = 0.0
        x0[:, 6] = 1.0
    return x0


def build_env(workload, handle, integrator, n_global, lo, hi, dev):
    from autonomous_quadrotor_environment_amd import BatchedQuad
    _, direct, controller, _, _, _, _ = WORKLOADS[workload]
    dt, compute, _ = HANDLES[handle]
    x0 = torch.from_numpy(np.ascontiguousarray(make_init(workload, n_global, lo, hi).T))
    if controller.startswith("ppo"):  # config 5 trains (solved/broken/timeout end episodes, 500-step episodes as ppo.py:301)
        env = BatchedQuad(hi - lo, 0.01, 500, training=(controller == "ppo_rollout"), direct_control=1, T=5,
                          integrator=integrator, device=dev, dtype=dt, compute=compute)
        env.reset(x0, return_history=False, push_window=True)
        return env
    env = BatchedQuad(hi - lo, 0.01, 10 ** 9, training=False, direct_control=direct, T=1,
                      integrator=integrator, device=dev, dtype=dt, compute=compute)
    env.reset(x0)
    if controller == "pid":
        env.pid_reset()
    return env


POLICY_FILE = os.path.join(ROOT, "tests", "golden", "g9_mlp_N128.npz")


def policy_arrays(policy):
    """(actor, critic) as lists [W1, b1, W2, b2, W3, b3] of float64 arrays.  trained: the reference's checked-in
    solved/nn_old_solved_128_32000_e8f2...pth re-exported as plain arrays by oracle/gen_golden.py (config 4's
    actor); random: nn.Linear's default init from numpy's RandomState(1) (ActorCriticWeights.random(128, seed=1))."""
    if policy == "trained":
        g = np.load(POLICY_FILE, allow_pickle=False)
        get = lambda head: [np.asarray(g[f"{head}_{l}_{p}"], dtype=np.float64) for l in (0, 2, 4) for p in ("weight", "bias")]
        return get("actor"), get("critic")
    rng = np.random.RandomState(1)

    def lin(o, i):
        b = 1.0 / np.sqrt(i)
        return rng.uniform(-b, b, (o, i)), rng.uniform(-b, b, o)

    def mlp(out):
        W1, b1 = lin(128, 75); W2, b2 = lin(128, 128); W3, b3 = lin(out, 128)
        return [W1, b1, W2, b2, W3, b3]
    return mlp(4), mlp(1)


def policy_label(policy):
    return {"trained": "trained (reference's checked-in N=128 actor / critic, re-exported: tests/golden/g9_mlp_N128.npz)",
            "random": "random(seed=1) (nn.Linear default init: the reset-heavy variant)"}[policy]


def load_policy(policy, dev):
    from autonomous_quadrotor_environment_amd import ActorCriticWeights
    a, c = policy_arrays(policy)
    w = ActorCriticWeights(a, c, dev)
    w.std = 0.1   # action_std of ppo.py:301 (FIXED_STD)
    return w


# qenv_ppo_rollout (in-band) / qenv_policy_rollout run policy + env step of a whole chunk as one launch (k_ppo_fused) unless
# QENV_PPO_FUSED=0 (A/B: the two launches per step)
FUSED_ROLLOUT = os.environ.get("QENV_PPO_FUSED", "1") != "0"


def make_ppo_runner(env, controller, chunk, use_graph, seed, env_offset, n_global, shards=1, make_env=None, n_local=0,
                    dev=None, policy="trained", inband=True):
    """config 4 / 5: the rollout collector of ppo.py:223-269 on the device."""
    from autonomous_quadrotor_environment_amd import PolicyRollout
    from autonomous_quadrotor_environment_amd import ActorCriticWeights
    from autonomous_quadrotor_environment_amd.distributed import all_gather_returns
    dev = env.device if env is not None else dev
    w = load_policy(policy, dev)
    if controller == "ppo_actor":
        ro = PolicyRollout(env, w.actor_only(), chunk)
        if FUSED_ROLLOUT:   # the evaluation loop as ONE library call (qenv_policy_rollout: one launch of k_ppo_fused per chunk)
            def fused_body():
                ro.t = 0
                ro.run_deterministic(chunk)
            return fused_body, chunk, ro

        def body():
            ro.t = 0
            for _ in range(chunk):
                ro.step(deterministic=True)
        if not use_graph:
            return body, chunk, ro
        assert chunk % env.T == 0, "a graph replay must bring the ring head back (chunk % T == 0)"
        side = torch.cuda.Stream(device=env.device)
        side.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(env.device).wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            body()
        return graph.replay, chunk, ro
    if shards > 1:  # the rank's range as independent shards on their own streams (StreamShardedRollout's docstring)
        from autonomous_quadrotor_environment_amd import StreamShardedRollout
        sro = StreamShardedRollout(make_env, n_local, w, chunk, shards=shards, seed=seed, env_offset=env_offset, device=dev,
                                   store_obs=True, auto_reset=True, inband=inband)

        def sharded_body():
            sro.t = 0
            sro.run(chunk)          # the whole horizon of every shard: one library call per shard (qenv_ppo_rollout)
            sro.finish(ddof=0, group=None)
            all_gather_returns(sro.last_returns(), n_global)
        return sharded_body, chunk, sro
    ro = PolicyRollout(env, w, chunk, store_obs=True, auto_reset=True, seed=seed, env_offset=env_offset, inband=inband)

    def body():  # one horizon: rollout, GAE (global normalisation), episode returns all-gathered over RCCL
        ro.t = 0
        ro.run(chunk)               # the whole horizon in one library call (qenv_ppo_rollout)
        ro.finish(ddof=0, group=None)
        all_gather_returns(env.last_return, n_global)
    return body, chunk, ro


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


# ---- timing ---------------------------------------------------------------------------------------------------------
def time_region(run, per_call, steps, world, dev):
    """barrier + sync | exactly `steps` env steps | sync + barrier; returns seconds (max over ranks).
    The opening bracket makes the ranks start together; a rank's time ends when ITS GPU has finished the K steps
    (torch.cuda.synchronize), and the job's time is the maximum over the ranks.  The closing barrier -- the MAX
    all-reduce itself -- comes after the clock is read: its own latency (tens of microseconds over RCCL, against a
    92 us region at the driver's K = 20) is not env work."""
    assert steps % per_call == 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // per_call):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # closes the bracket: every rank has finished its K steps
        torch.cuda.synchronize()
        dt = float(t.item())
    return dt


def timed_regions(run, per_call, steps, world, dev, min_seconds, max_regions=5000):
    """The K-step region repeated until `min_seconds` have been timed (at least once): the list of region times.
    Every region is bracketed as the contract says; the times are max-over-ranks, so all ranks agree on the count."""
    ts, total = [], 0.0
    while True:
        dt = time_region(run, per_call, steps, world, dev)
        ts.append(dt)
        total += dt
        if total >= min_seconds or len(ts) >= max_regions:
            return ts


def kernel_launch_us(run, per_call, launch, reps=8, train=0):
    """Average launch-to-launch interval (kernel duration + dependent-launch boundary) of the dominant kernel on a
    saturated stream, from HIP events on the launch stream (the env launches on
    torch's current stream, so torch.cuda.Event brackets it).  An event pair costs more than a short kernel (its two
    marker packets added ~20 us to a 70 us launch when every launch had its own pair), so one pair brackets a train of
    back-to-back launches -- at least 2 ms of them -- and the elapsed time is divided by the number of launches: the
    average launch duration in the steady state, including the dependent launch boundary (~1 us), which makes
    `achieved` conservative; rocprofv3 --kernel-trace of the same command gives the same figure (profiles/).
    launch='step': `run` is one hipGraph replay of `per_call` launches, the result is divided by per_call as well."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(); run(); b.record()
    torch.cuda.synchronize()
    whole = train > 0   # a train of a fixed length (one whole episode of an episodic workload: every phase of the episode
    #                     -- the retry-heavy first steps, the reset launch -- is in every bracket): the plain mean
    train = train or int(min(2000, max(4, np.ceil(2000.0 / max(a.elapsed_time(b) * 1e3, 1.0)))))
    ts = []
    for _ in range(reps):
        a.record()
        for _ in range(train):
            run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / train)
    ts.sort()
    us = float(np.mean(ts)) if whole else float(np.mean(ts[: max(1, len(ts) * 3 // 4)]))  # drop the slowest quarter (clock ramps, interrupts)
    return us if launch == "fused" else us / per_call


def recorded(name):
    """A figure recorded by an earlier profiling run (profiles/*.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def pick(total, cap, mult=1):
    ok = [c for c in range(1, cap + 1) if total % c == 0 and c % mult == 0]
    return max(ok) if ok else 0


def measure(workload, handle, integrator, launch, steps, warmup, chunk_arg, no_graph, envs, rank, world, dev,
            min_seconds, stream_shards=1, policy="trained", schedule="inband", want_sustained=False, want_step_api=False,
            placement="regroup"):
    """One workload on this rank's shard: W untimed warm-up steps, the K-step region repeated (median), the dominant
    kernel's per-launch time by HIP events, and its roofline block.  Returns (result dict, env, runner info)."""
    from autonomous_quadrotor_environment_amd.distributed import all_gather_returns, shard_range
    envs_per_gpu, direct, controller, flags, bytes_per_step, _, desc = WORKLOADS[workload]
    envs_per_gpu = envs or envs_per_gpu
    n_global = envs_per_gpu * world
    lo, hi = shard_range(n_global, rank, world)
    ppo = controller.startswith("ppo")
    # steps per call (fused launch / graph replay / rollout horizon): the largest divisor of the K timed steps, so the
    # timed region is independent of W; the W warm-up steps run first through their own (same-kind) runner when W is not
    # a multiple of it
    cap = 128 if controller == "ppo_rollout" else 250
    fused_ppo = ppo and FUSED_ROLLOUT and integrator == "rk45"
    mult = 5 if (controller == "ppo_actor" and not no_graph and not fused_ppo) else 1  # a graph replay must bring the ring head back
    chunk = chunk_arg or pick(steps, cap, mult)
    if not chunk:  # K has no multiple-of-5 divisor: eager launches instead of a graph replay
        no_graph, mult = True, 1
        chunk = pick(steps, cap)
    if steps % chunk:
        raise SystemExit("--chunk must divide --steps")
    shards = stream_shards if controller == "ppo_rollout" else 1
    shard_envs = {}

    def make_env(a, b):  # one handle per local shard range, shared by the warm-up runner and the timed one
        if (a, b) not in shard_envs:
            shard_envs[(a, b)] = build_env(workload, handle, integrator, n_global, lo + a, lo + b, dev)
        return shard_envs[(a, b)]
    env = None if shards > 1 else build_env(workload, handle, integrator, n_global, lo, hi, dev)
    if ppo:
        launch = "step"

    def runner(c, graph_ok=True):
        if ppo:
            return make_ppo_runner(env, controller, c, graph_ok and (not no_graph) and controller == "ppo_actor",
                                   seed=4, env_offset=lo, n_global=n_global, shards=shards, make_env=make_env,
                                   n_local=hi - lo, dev=dev, policy=policy, inband=(schedule == "inband"))
        return make_runner(env, controller, flags, launch, c, not no_graph) + (None,)
    # Episodes: the closed-loop workloads without an auto-reset (LQR / PID rollouts, actor in the loop) run EPISODES of 500
    # steps like the reference's scripts (lqr_quad.py / ppo_quad_eval.py run one episode per reset); without this the
    # repeated K-step regions would carry the same envs for thousands of steps, where the few that the controller (or the
    # random-init policy) does not stabilise need ever more RK45 sub-steps and every wave waits for them.  The reset to
    # the initial states is part of the timed work (one launch per 500 steps).
    episodic = controller in ("lqr", "pid", "ppo_actor")
    x0_dev = torch.from_numpy(np.ascontiguousarray(make_init(workload, n_global, lo, hi).T)).to(dev) if episodic else None
    ep = {"n": 0}

    def episodic_run(inner, pc):
        if not episodic:
            return inner

        def wrapped():
            if ep["n"] + pc > EPISODE_STEPS:
                if controller == "ppo_actor":
                    env.reset(x0_dev, return_history=False, push_window=True)
                else:
                    env.reset(x0_dev)
                    if controller == "pid":
                        env.pid_reset()
                ep["n"] = 0
            inner()
            ep["n"] += pc
        return wrapped
    # Placement (BatchedQuad.regroup): a wave runs as many RK45 attempts per env step as its slowest lane needs, and ~5 % of the
    # random initial states are never stabilised by the LQR (they tumble and take two or three sub-steps per env step for the
    # rest of the episode) -- in the natural order nearly every wave of 64 holds one.  One UNTIMED placement episode from the
    # workload's own initial states, then the envs are sorted by the body-rate norm they ended with; every timed episode
    # starts from the same initial states, in that order.  Per-env results are bit-identical (tests/test_gpu_parity.py);
    # the returns are gathered in the caller's order (env.logical).  --placement natural: the batch as drawn.
    placed = None
    # (Config 4, the trained actor in the loop, was measured with the same placement: 2.80 against 2.85 G in the order drawn --
    # the policy stabilises 99 % of the envs, their extra sub-steps are the first steps' transients, and |omega| at the
    # episode's end predicts nothing: left as drawn.)
    if placement == "regroup" and episodic and launch == "fused" and controller in ("lqr", "pid"):
        t0 = time.perf_counter()
        for _ in range(EPISODE_STEPS // 250):
            env.rollout(controller, 250, flags=flags)
        order = env.regroup()   # every per-env row of the handle moves with its env (qenv_permute_envs); the hard envs first
        env.reset(x0_dev)
        if controller == "pid":
            env.pid_reset()
        torch.cuda.synchronize()
        placed = {"policy": "BatchedQuad.regroup(): envs sorted by |omega| at the end of one untimed 500-step placement episode "
                            "from the workload's own initial states (qenv_permute_envs moves every per-env row); per-env trajectories bit-identical, "
                            "outputs returned in the caller's order (env.logical)",
                  "setup_seconds": time.perf_counter() - t0,
                  "moved_envs": int((order != torch.arange(order.numel(), device=order.device)).sum())}
    run, per_call, ro = runner(chunk)
    run = episodic_run(run, per_call)
    if warmup and warmup % chunk == 0:
        time_region(run, per_call, warmup, world, dev)     # W untimed warm-up steps
    elif warmup:
        cw = pick(warmup, cap, mult)
        rw, pcw, _ = runner(cw, True) if cw else runner(pick(warmup, cap), False)  # no %T divisor: eager warm-up
        time_region(episodic_run(rw, pcw), pcw, warmup, world, dev)
    regions = timed_regions(run, per_call, steps, world, dev, min_seconds)   # each: exactly K timed steps
    # An episodic workload's K-step region costs what its phase of the episode costs (the first ~100 steps after a reset carry
    # the controllers' transients: more RK45 sub-steps; with the envs regrouped the later phases are cheaper still), so the
    # median region is a late-episode one: the MEAN over regions that cover whole episodes is the episode's average rate.
    # Other workloads: the median (robust against a stray slow region).
    whole_episodes = episodic and len(regions) * steps >= EPISODE_STEPS
    secs = float(np.mean(regions)) if whole_episodes else float(np.median(regions))
    statistic = ("mean of the K-step regions (each barrier+sync bracketed), which cover %.1f episodes of %d steps" % (len(regions) * steps / EPISODE_STEPS, EPISODE_STEPS)
                 if whole_episodes else "median of the K-step regions (each barrier+sync bracketed)")
    value = n_global * steps / secs
    envs_run = [env] if shards == 1 else list(shard_envs.values())
    finite = all(bool(torch.isfinite(e.state).all()) for e in envs_run)
    # the path's one exchange step (RCCL all-gather over xGMI of per-env episode returns, ppo.py:356-382's Pool gather):
    # every workload that has rewards gathers once more after the timed region, timed by itself (config 5 also gathers
    # inside every timed horizon)
    gathered, gather_us = None, None
    if flags != 1:
        local = (torch.cat([e.last_return for e in envs_run]) if controller == "ppo_rollout"
                 else env.logical(env.episode_return))
        all_gather_returns(local, n_global)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        gathered = all_gather_returns(local, n_global)
        torch.cuda.synchronize()
        gather_us = (time.perf_counter() - t0) * 1e6

    # dominant kernel, per-launch duration by HIP events on the launch stream
    spl = chunk if launch == "fused" else 1
    # (the PPO workloads time their dominant kernel, the policy forward, by itself below: `run` is a whole horizon there,
    # collectives included)
    if ppo:
        k_us = 1.0
    elif episodic and launch == "fused":   # whole episodes per event bracket (the reset launch rides along: ~10 us in 15 ms)
        ep["n"] = EPISODE_STEPS            # every bracket starts at an episode start
        k_us = kernel_launch_us(run, per_call, launch, reps=4, train=max(1, EPISODE_STEPS // per_call))
    else:
        k_us = kernel_launch_us(run, per_call, launch)
    algo_bytes = bytes_per_step * (hi - lo) * spl
    achieved = algo_bytes / (k_us * 1e-6) / 1e9
    ttype = {"f32": "float, float", "mix": "float, double", "lean": "double, double (lean)", "f64": "double, double"}[handle]
    kname = ("k_rollout" if (launch == "fused" or controller != "const") else "k_step") + f"<{ttype}, {integrator}, {controller}>"
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "kernel": kname,
                "kernel_us": k_us, "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_per_env_step": bytes_per_step, "env_steps_per_launch": (hi - lo) * spl}
    if launch == "fused":
        roofline["achieved_is"] = ("algorithmic bytes (SURVEY 8d figure x env steps per launch) / launch time; the fused "
                                   "kernel keeps the state in registers between steps, the HBM bytes it really moves are "
                                   "`traffic` -- its limiter is VALU issue (`valu`), not HBM")
        roofline["limiter"] = "valu_issue"
    tr = recorded("traffic.json")
    key = f"{workload}:{integrator}:{launch}:{handle}" + (":regroup" if placed else "")   # (the rows that move are the same in either order)
    if tr and key in tr:
        roofline["traffic"] = tr[key]
        roofline["traffic_unit"] = "B/launch"
        roofline["traffic_source"] = "recorded: profiles/traffic.json (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, not measured in this run)"
    va = recorded("valu.json")
    if va and key in va:  # SQ counters of the same kernel, recorded by tools/pmc_valu_count.sh
        v = va[key]
        spw = (hi - lo) * spl / 64.0 / (k_us * 1e-6)   # wave-steps per second of this rank
        roofline["valu"] = {"source": "recorded: profiles/valu.json (" + v.get("source", "rocprofv3 PMC") + ")",
                            "instructions_per_env_step_per_wave": v["valu_per_wave_step"],
                            "issue_cycles_per_instruction": v["active_valu_cycles_per_inst"], "simds": 1024,
                            "clock_ghz": v["clock_ghz"],
                            "issue_frac": spw * v["valu_per_wave_step"] * v["active_valu_cycles_per_inst"] / (1024 * v["clock_ghz"] * 1e9)}
    if ppo:  # the dominant kernel of the PPO workloads is the policy MLP forward on the matrix cores
        heads = 2 if controller == "ppo_rollout" else 1
        e0, r0 = (env, ro) if shards == 1 else (ro.ros[0].env, ro.ros[0])   # one launch = one shard's envs
        n_l = e0.n_envs
        if fused_ppo:   # ONE launch carries `chunk` steps of policy + env step for this shard's envs (k_ppo_fused)
            def one_launch():
                r0.t = 0
                r0.run_deterministic(chunk) if controller == "ppo_actor" else r0.run(chunk)
            mlp_us = kernel_launch_us(one_launch, 1, "fused")
            if controller == "ppo_actor":   # (the timed launches moved the envs on: back to an episode start)
                env.reset(x0_dev, return_history=False, push_window=True)
                ep["n"] = 0
            tiles = ((n_l + 31) // 32) * chunk
        else:
            mlp_us = kernel_launch_us(lambda: e0.mlp_forward(r0.w, "actor", out=r0.mean), 1, "fused")
            tiles = (n_l + 31) // 32
        t_s = mlp_us * 1e-6
        cyc_tile = sum(n * c for n, c, _ in MLP_MFMA.values())
        flop_tile = sum(n * f for n, _, f in MLP_MFMA.values())
        busy = tiles * cyc_tile / (1024 * MAX_CLOCK_GHZ * 1e9 * t_s)
        executed = tiles * flop_tile / t_s / 1e12
        tfl = MLP_FLOPS * n_l * (chunk if fused_ppo else 1) / t_s / 1e12
        roofline = {"bound": "mfma", "achieved": executed, "peak": executed / busy, "unit": "TFLOP/s", "frac": busy,
                    "traffic": None,
                    "kernel": ("k_ppo_fused (per step and 64-env block: actor forward 75-128-128-4 of two 32-env tiles, then the env "
                               "step; one launch per %d steps)" % chunk) if fused_ppo else "k_mlp_v3 (actor forward 75-128-128-4)",
                    "kernel_us": mlp_us, "env_steps_per_launch": n_l * (chunk if fused_ppo else 1),
                    "kernel_us_per_step": mlp_us / (chunk if fused_ppo else 1),
                    "envs_per_launch": n_l, "mlp_forwards_per_env_step": heads,
                    "achieved_is": "EXECUTED matrix-core flops per launch (the instruction mix below, padded tiles included) / "
                                   "launch time; `peak` = that same mix with the matrix pipe of all 1024 SIMDs busy every cycle "
                                   f"at {MAX_CLOCK_GHZ} GHz, so `frac` = matrix-pipe busy fraction (at the nominal clock: a lower "
                                   "bound of the busy fraction at the clock the chip really holds)",
                    "mfma_per_32env_tile": {k: {"count": n, "pipe_cycles_each": c} for k, (n, c, _) in MLP_MFMA.items()},
                    "mfma_cycles_per_tile": cyc_tile,
                    "bf16_products_tflops": MLP_MFMA["v_mfma_f32_32x32x16_bf16"][0] * MLP_MFMA["v_mfma_f32_32x32x16_bf16"][2]
                                            * tiles / t_s / 1e12,
                    "bf16_dense_peak_tflops": 2516.0,
                    "algorithmic": {"fp32_tflops": tfl, "flops_per_env": MLP_FLOPS,
                                    "note": "53.0 kflop per env and forward (SURVEY 8a a20) / launch time; NOT a roofline "
                                            "fraction of anything: layers 1-2 run as 6 bf16 products of 3-way split fp32 "
                                            "operands (fp32-accurate), i.e. ~6x the algorithmic flops on a 16x faster pipe"},
                    "hbm_algorithmic_GBps_whole_step": bytes_per_step * value / world / 1e9}
        st = recorded("r03_fused_stamps_sampling.json" if controller == "ppo_rollout" else "r03_fused_stamps_deterministic.json")
        if fused_ppo and st:   # in-kernel s_memtime stamps of the same launch (tests/tools/fused_stamps.py), recorded
            ph = st["phases_p10_p50_p90"]
            tile = ph["A: layers 1-3"][1]
            roofline["policy_tile"] = {"cycles_median": tile, "mfma_cycles": cyc_tile, "matrix_pipe_busy_in_tile": cyc_tile / tile,
                                       "block_step_cycles_median": st["step_cycles_median"], "env_step_cycles_median": ph["env step"][1],
                                       "source": "recorded: profiles/r03_fused_stamps_*.json (a 32-env tile of layers 1-3 on its lone "
                                                 "wave; a block-step = two tiles + the env step of 64 envs)"}
    # what the same kernel sustains when the launch granularity is not tied to K: 250-step launches, one bracket around
    # >= min_seconds of them (the driver's K = 20 makes every region a single 20-step launch + its sync); episodic
    # workloads keep their 500-step episodes (two launches and a reset each)
    sustained = None
    if want_sustained and not ppo and launch == "fused":
        r2, pc, _ = runner(250)
        r2 = episodic_run(r2, pc)
        ep["n"] = EPISODE_STEPS
        time_region(r2, pc, 500, world, dev)
        t1 = time_region(r2, pc, 500, world, dev)
        n_l = max(1, int(np.ceil(min_seconds / max(t1, 1e-6))))
        if world > 1:  # the same launch count on every rank
            t = torch.tensor([n_l], device=dev, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_l = int(t.item())
        s_ = time_region(r2, pc, 500 * n_l, world, dev)
        sustained = {"value": n_global * 500 * n_l / s_, "unit": "env-steps/s", "steps": 500 * n_l, "steps_per_launch": 250,
                     "seconds": s_, "hbm_frac": n_global / world * 500 * n_l / s_ * bytes_per_step / 1e9 / HBM_PEAK_GBPS}
    step_api = None
    if want_step_api and not ppo and world == 1:   # the path quad.step() callers take: one launch per env step (hipGraph replay)
        if placed is not None:   # per-env actions from the caller: the batch in its natural order
            env.clear_placement()
        r3, pc3, _ = (make_runner(env, controller, flags, "step", 250, not no_graph) + (None,))
        r3 = episodic_run(r3, pc3)
        ep["n"] = EPISODE_STEPS
        time_region(r3, pc3, 250, world, dev)
        step_api = {"value": n_global * 1000 / time_region(r3, pc3, 1000, world, dev), "unit": "env-steps/s",
                    "launch": "one kernel launch per env step, 250 per hipGraph replay" if not no_graph else "one eager launch per env step"}
    res = {"value": value, "ms_per_step": secs / steps * 1e3, "regions": len(regions), "statistic": statistic,
           "ms_per_step_median": float(np.median(regions)) / steps * 1e3,
           "ms_per_step_mean": float(np.mean(regions)) / steps * 1e3, "sustained": sustained, "step_api": step_api,
           "ms_per_step_min": min(regions) / steps * 1e3, "ms_per_step_max": max(regions) / steps * 1e3,
           "timed_seconds": float(sum(regions)), "finite": finite, "chunk": chunk, "launch": launch,
           "hip_graph": (launch == "step" and not no_graph and controller != "ppo_rollout" and not fused_ppo),
           "policy_env_fused": fused_ppo,
           "envs_per_gpu": hi - lo, "global_envs": n_global, "controller": controller, "desc": desc, "roofline": roofline,
           "all_gather_len": None if gathered is None else int(gathered.numel()), "all_gather_us": gather_us,
           "stream_shards": shards, "episode_steps": EPISODE_STEPS if episodic else None,
           "weights": (policy_label(policy) if ppo else None),
           "schedule": (schedule if controller == "ppo_rollout" else None), "env_placement": placed}
    if controller == "ppo_rollout" and schedule == "inband":   # share of the rollout rows that are policy steps (the others are
        ros = [ro] if shards == 1 else ro.ros                   # the restarted envs' warm-up steps: env steps, but no samples)
        res["obs_layout"] = "slices" if all(getattr(r, "slice_obs", False) for r in ros) else "windows"
        res["valid_row_fraction"] = float(sum(float(r.valid[:r.t].sum()) for r in ros) / max(1, sum(r.valid[:r.t].numel() for r in ros)))
    return res, env, (controller, flags, no_graph, bytes_per_step, n_global)


def cpu_baseline(workload, integrator, seconds, policy="trained", steps=1000):
    """The C oracle (a port of the reference's algorithm, pinned by the reference's golden vectors) timed on this host's
    cores on a bounded sample of the same workload: the fp64 build on one thread and on all the cores
    this process may use, and the fp32 build of the same source on all cores.  `seconds` bounds the whole call."""
    from oracle import quad_oracle as qo
    qo.build()
    _, direct, controller, _, _, _, _ = WORKLOADS[workload]
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the GPU box's CPU share for one GPU is 16
    x0 = make_init(workload, 4096 * threads, 0, 4096 * threads)
    if controller.startswith("ppo"):  # actor (+ critic for config 5) MLP in the loop, the same weights as the GPU run
        actor, critic = policy_arrays(policy)
        batch = lambda c, n, f32=False: qo.batch_ppo(c, n, actor, critic if controller == "ppo_rollout" else None, T=5,
                                                     integrator=integrator, fp32=f32)
    else:
        ctrl = {"const": 0, "lqr": 1, "pid": 2}[controller]
        batch = lambda c, n, f32=False: qo.batch_rollout(c, n, direct=direct, integrator=integrator, controller=ctrl,
                                                         fp32=f32)
    # calibrate on one thread (this IS the single-thread figure), then size the samples: ~40 % of the budget for fp64 on
    # all cores, ~30 % for fp32 on all cores
    n1 = 64 if seconds >= 4 else 16
    t0 = time.perf_counter()
    batch(x0[:n1], 50)
    rate1 = n1 * 50 / (time.perf_counter() - t0)

    def all_cores(f32, budget):
        per_thread = int(max(8, min(4096, rate1 * budget / steps)))
        chunks = [x0[k * per_thread:(k + 1) * per_thread] for k in range(threads)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:  # ctypes releases the GIL inside the C call
            list(ex.map(lambda c: batch(c, steps, f32), chunks))
        dt = time.perf_counter() - t0
        return threads * per_thread * steps / dt, per_thread, dt
    v64, per64, dt64 = all_cores(False, 0.4 * seconds)
    v32, per32, dt32 = all_cores(True, 0.3 * seconds)
    return {"value": v64, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{threads} threads x {per64} envs x {steps} steps of {workload} ({integrator}), fp64 build",
            "what": "fp64 C restatement of the reference (the oracle/ directory), pinned by reference-generated fixtures",
            "threads": threads, "envs_per_thread": per64, "steps": steps, "seconds": dt64,
            "single_thread_rate": rate1,
            "fp32_build": {"value": v32, "cores": threads, "seconds": dt32,
                           "sample": f"{threads} threads x {per32} envs x {steps} steps, float build of the same source",
                           "what": "the float build of the same C source (oracle/: double read as float): float storage and arithmetic (timing baseline only; parity is "
                                   "judged on the fp64 build)"},
            "weights": policy_label(policy) if controller.startswith("ppo") else None,
            "reference_python_rate": 800.0,
            "reference_python_rate_source": "the reference's own Python step, measured in the survey container per core "
                                            "(BASELINE.md section 3); it cannot travel to the GPU box"}


def next_rows(dev, policy):
    """Dense tail of the landing policy at config 4's batch, and one learner epoch at config 5's horizon (a few hundred ms)."""
    import autonomous_quadrotor_environment_amd as aq
    out = {}
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *sh: torch.randn(*sh, device=dev, generator=g)
    K, H, n_out, N = 800, 64, 3, 131072
    W = [r(H, K) / K ** 0.5, r(H) * 0.1, r(H, H) / H ** 0.5, r(H) * 0.1, r(n_out, H) / H ** 0.5, r(n_out) * 0.1]
    x = torch.rand(K, N, device=dev, generator=g) * 2 - 1
    y = torch.empty(n_out, N, device=dev)
    us = kernel_launch_us(lambda: aq.dense_tail(x, W, final_tanh=True, out=y), 1, "fused", reps=4)
    flops = 2.0 * N * (K * H + H * H + H * n_out)
    mfma3 = (K // 16 + (1 if K % 16 else 0) + 4) * 2 * 6     # split-bf16 32x32x16 MFMAs per 32-sample tile: layers 1 and 2, two output tiles
    out["dense_tail_800_64_64_3"] = {"samples": N, "kernel": "k_dense_tail_v3 (split-bf16 matrix cores, fp32-accurate; W1 streamed as fragments)",
                                     "kernel_us": us, "samples_per_s": N / us * 1e6,
                                     "bound": "the CU's fill rate (~10 B per clock) while it multiplies, nothing requested in the split / barrier phases "
                                              "(profiles/r05_dense_tail_stamps.json: the next chunk's loads go out k-step by k-step between the products; "
                                              "in one burst behind the chunk barrier they took the last waves 5 k cycles to get accepted: 135 us); "
                                              "HBM traffic = the features once",
                                     "algorithmic_TFLOPs": flops / us / 1e6,
                                     "mfma_pipe_busy_at_nominal_clock": mfma3 * 32 * (N / 32) / 1024 / (us * 1e-6 * MAX_CLOCK_GHZ * 1e9),
                                     "hbm_GBps": (K + n_out) * 4 * N / us / 1e3, "hbm_frac": (K + n_out) * 4 * N / us / 1e3 / HBM_PEAK_GBPS,
                                     "data": "synthetic U(-1,1) features"}
    del x, y
    n, S = 65536, 128
    w = load_policy(policy, dev)
    env = aq.BatchedQuad(n, 0.01, 500, training=True, direct_control=1, T=5, device=dev)
    ro = aq.PolicyRollout(env, w, S, store_obs=True, auto_reset=True, seed=3, inband=True)
    ro.start()
    ro.run(S)
    ret, adv = ro.finish(ddof=0, group=False)
    learner = aq.PpoLearner(w, K_epochs=1, group=False)
    learner.update(ro, ret, adv)      # buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        learner.update(ro, ret, adv)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    samples = n * S
    fl = samples * 2 * (75 * 128 + 128 * 128 + 128 * 4 + 75 * 128 + 128 * 128 + 128) * 3   # forward + 2x backward, both heads
    out["ppo_update_epoch"] = {"samples": samples, "valid_samples": int(float(ro.valid[:S].sum())), "seconds": dt,
                               "samples_per_s": samples / dt,
                               "kernels": "k_mlp_grad_v5 x 2 (one launch per head: forward recompute, fused loss, backward AND the six weight / "
                                          "bias gradients; the workgroup's eight waves -- two per SIMD, 16 hidden rows each on 16x16x32 bf16 "
                                          "tiles -- share a 32-sample tile, activations exchanged through LDS images, transposed by "
                                          "ds_read_b64_tr_b16), k_adam",
                               "algorithmic_TFLOPs": fl / dt / 1e12,
                               # the samples themselves, read once per head: observations 75 floats + actions 4 + old log-probs 4 + adv +
                               # valid (actor) / returns + valid (critic); the counters agree (profiles/r05_update_epoch_counters.json:
                               # FETCH_SIZE x 2 = 2.7 GB, WRITE_SIZE 0.07 GB per launch; rounds 1-4: 37.3 GB per head)
                               "hbm_GB_per_epoch": samples * 4 * (2 * 75 + 4 + 4 + 1 + 1 + 1 + 1) / 1e9,
                               "hbm_GBps": samples * 4 * (2 * 75 + 4 + 4 + 1 + 1 + 1 + 1) / dt / 1e9,
                               "hbm_GB_per_epoch_rounds_1_to_4": 2 * 37.3,
                               "bound": "vector issue + matrix pipe + LDS in turn (two waves per SIMD in step behind five barriers per tile: ~600 "
                                        "vector, 152 matrix and 187 LDS instructions per wave and tile; matrix pipe 0.36, LDS 0.60 busy; "
                                        "profiles/r05_grad5_stamps.json, r05_update_epoch_counters.json); HBM traffic is the samples, once",
                               "weights": policy_label(policy), "schedule": "inband"}
    return out


def slice_of(r, name, handle):
    """A `configs` entry from a measure() result."""
    rf = r["roofline"]
    out = {"value": r["value"], "unit": "env-steps/s (whole job)" if r["global_envs"] != r["envs_per_gpu"]
           else "env-steps/s (one GPU's slice)", "ms_per_step": r["ms_per_step"],
           "regions": r["regions"], "envs_per_gpu": r["envs_per_gpu"], "global_envs": r["global_envs"], "handle": handle,
           "workload": r["desc"], "weights": r["weights"], "finite": r["finite"], "dominant_kernel": rf["kernel"],
           "kernel_us": rf["kernel_us"], "bound": rf["bound"], "frac": rf["frac"],
           "achieved": rf["achieved"], "roofline_unit": rf["unit"], "stream_shards": r["stream_shards"],
           "episode_steps": r["episode_steps"], "envs_per_kernel_launch": rf.get("envs_per_launch", r["envs_per_gpu"]),
           "all_gathered_returns": r["all_gather_len"], "all_gather_us": r["all_gather_us"]}
    for k in ("schedule", "valid_row_fraction", "obs_layout", "sustained", "step_api"):
        if r.get(k) is not None:
            out[k] = r[k]
    for k in ("valu", "mfma_cycles_per_tile", "algorithmic", "bf16_products_tflops", "traffic"):
        if k in rf:
            out[k] = rf[k]
    return out


COMPACT_KEYS = ("value", "ms_per_step", "steps", "envs_per_gpu", "global_envs", "handle", "finite", "bound", "frac", "kernel_us",
                "all_gathered_returns", "all_gather_us", "valid_row_fraction")


def compact(sl):
    """The few fields of a slice that the contract line repeats (the whole slice is in the detail line)."""
    out = {k: sl[k] for k in COMPACT_KEYS if sl.get(k) is not None}
    if sl.get("weights"):
        out["weights"] = sl["weights"].split(" ")[0]
    if sl.get("sustained"):
        out["sustained_value"] = sl["sustained"]["value"]
    if isinstance(sl.get("valu"), dict):
        out["valu_issue_frac"] = sl["valu"]["issue_frac"]
    if isinstance(sl.get("cpu_baseline"), dict):
        out["cpu_value"] = sl["cpu_baseline"]["value"]
    return out


def measured_row_copy(dev):
    """The HBM denominator measured on this box: a row copy with the env kernels' access shape (13 SoA rows,
    one lane per env), 872 MB read + 872 MB written per launch, far beyond the 256 MiB Infinity Cache.  GB/s."""
    from autonomous_quadrotor_environment_amd import _lib as L
    import ctypes as C
    n_c = 16 * 1024 * 1024
    src = torch.rand(13, n_c, device=dev, dtype=torch.float32); dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    copy = lambda: L.check(L.lib.qenv_debug_copy_rows(L.F32, L.ptr(dst), L.ptr(src), 13, n_c, st), "copy")
    for _ in range(3):
        copy()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a_.record()
    for _ in range(10):
        copy()
    b_.record(); torch.cuda.synchronize()
    return round(2 * 13 * n_c * 4 * 10 / (a_.elapsed_time(b_) * 1e-3) / 1e9, 1)


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        self_launch(args)  # never returns
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    from autonomous_quadrotor_environment_amd.distributed import init_process_group_from_env
    rank, world, local = init_process_group_from_env()
    assert world == args.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    handle = WORKLOADS[args.workload][5] if args.dtype == "auto" else args.dtype
    default_run = args.workload == DEFAULT_WORKLOAD and args.envs == 0   # the line the driver records

    res, env, (controller, flags, no_graph, bytes_per_step, n_global) = measure(
        args.workload, handle, args.integrator, args.launch, args.steps, args.warmup, args.chunk, args.no_graph,
        args.envs, rank, world, dev, args.min_seconds, args.stream_shards, args.policy, args.schedule,
        want_sustained=not args.no_sustained, want_step_api=default_run and not args.no_extras, placement=args.placement)
    ppo = controller.startswith("ppo")
    qualified = WORKLOADS[args.workload][5]
    rf = res["roofline"]
    if "valu" in rf:   # both fractions side by side: HBM on the algorithmic bytes (the contract's `frac`) and vector issue
        rf["hbm_algorithmic_frac"] = rf["frac"]
        rf["valu_issue_frac"] = rf["valu"]["issue_frac"]
    out = {
        "metric": "quadrotor env-steps/sec", "value": res["value"], "unit": "env-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": HANDLES[handle][2], "data": "synthetic",
        "config": {"workload": res["desc"], "envs_per_gpu": res["envs_per_gpu"], "global_envs": res["global_envs"],
                   "integrator": args.integrator, "launch": res["launch"], "controller": controller,
                   "handle": handle,
                   "handle_is": {"f32": "fp32 rows, fp32 arithmetic", "mix": "fp64 rows, fp32 arithmetic",
                                 "lean": "fp64 rows, lean fp64 arithmetic", "f64": "fp64 rows, IEEE fp64 arithmetic"}[handle],
                   "parity_qualified": handle in (qualified, "f64"),
                   "steps_per_graph_replay_or_launch": res["chunk"], "hip_graph": res["hip_graph"],
                   "parallelism": f"env-sharded x{world}", "stream_shards_per_gpu": res["stream_shards"],
                   "episode_steps": res["episode_steps"],
                   "finite": res["finite"], "weights": res["weights"], "schedule": res.get("schedule"),
                   "valid_row_fraction": res.get("valid_row_fraction"), "obs_layout": res.get("obs_layout"),
                   "all_gathered_returns": res["all_gather_len"], "all_gather_us": res["all_gather_us"],
                   "env_placement": res.get("env_placement") or "natural order"},
        "timing": {"regions": res["regions"], "statistic": res["statistic"], "ms_per_step_median": res["ms_per_step_median"],
                   "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"],
                   "ms_per_step_mean": res["ms_per_step_mean"], "timed_seconds": res["timed_seconds"]},
        "roofline": rf,
    }
    if default_run:
        out["headline_changed"] = ("config2->config3 (round 5): `value` is BASELINE configs[2], the largest single-GPU "
                                   "configuration; rounds 1-4 quoted configs[1], now configs.config2 / configs.config2_hover")
    if res["sustained"]:
        out["sustained"] = res["sustained"]
    if res["step_api"]:
        out["per_step_api"] = res["step_api"]
    del env
    detail = {}

    # --extras: the other operating points of a non-PPO workload (a few seconds): the per-step API path (hipGraph-replayed
    # qenv_step launches) and the fixed-step integrators, each in both launch modes, and the fp32 handle
    if args.extras and world == 1 and not ppo:
        from autonomous_quadrotor_environment_amd.distributed import shard_range
        lo, hi = shard_range(n_global, rank, world)
        extras = {}
        ck = 250
        for integ in ("rk45", "dp5", "rk4"):
            e2 = build_env(args.workload, handle, integ, n_global, lo, hi, dev)
            for launch in ("step", "fused"):
                r2, pc = make_runner(e2, controller, flags, launch, ck, not args.no_graph)
                time_region(r2, pc, ck, 1, dev)
                s = time_region(r2, pc, 500, 1, dev)
                extras[f"{integ}:{launch}"] = n_global * 500 / s
            del e2
        if handle != "f32":  # the fp32 handle (outside the tolerance over >= 500 open-loop steps) for comparison
            e2 = build_env(args.workload, "f32", "rk45", n_global, lo, hi, dev)
            r2, pc = make_runner(e2, controller, flags, "fused", ck, False)
            time_region(r2, pc, ck, 1, dev)
            extras["rk45:fused:f32_handle"] = n_global * 500 / time_region(r2, pc, 500, 1, dev)
            del e2
        detail["extras_env_steps_per_s"] = extras
        detail["extras_note"] = "500-step regions from the initial states, 250-step launches / graph replays, same handle as `value` unless named"
        # the same operating points as a fraction of the 8 TB/s HBM peak on the algorithmic bytes per env step;
        # BASELINE.json words config 2 as "RK4 dynamics only": that operating point is extras["rk4:fused"] of --workload config2
        detail["extras_hbm_frac"] = {k: round(v * bytes_per_step / 1e9 / HBM_PEAK_GBPS, 4) for k, v in extras.items()}
    if default_run and world == 1 and not args.no_extras:
        out["roofline"]["measured_row_copy_GBps"] = measured_row_copy(dev)
        out["roofline"]["frac_of_measured_row_copy"] = rf["achieved"] / out["roofline"]["measured_row_copy_GBps"]

    # short per-GPU slices of the other BASELINE configs, each on its qualified handle: the driver's record then covers the
    # dynamics-only step, the PPO actor in the loop and the full PPO rollout + GAE besides the LQR headline.  One GPU:
    # config 2 (perturbed hover, and literally at hover), config 4 (trained actor) and its random-policy variant, config 5
    # (trained and random policy), each with a ~1 s CPU baseline beside it.  N > 1: BASELINE defines configs 4 and 5 on the
    # whole node -- they run at their per-GPU sizes on every rank (max-over-ranks timing, like `value`) and end with the
    # RCCL all-gather of the episode returns, whose length and own time are in the slice.
    if default_run and not args.no_subconfigs:
        subs = {}
        if world == 1:
            plan = (("config2", "config2", 20, 20, None), ("config2_hover", "config2_hover", 250, 250, None),
                    ("config4", "config4", 250, 50, "trained"),
                    ("config4_random_policy", "config4", 250, 50, "random"), ("config5", "config5", 128, 128, "trained"),
                    ("config5_random_policy", "config5", 128, 128, "random"))
        else:
            plan = (("config4", "config4", 250, 50, "trained"), ("config5", "config5", 128, 128, "trained"))
        if world == 1 and res.get("env_placement"):   # the headline's workload with the batch as drawn, same K
            plan = (("config3_natural_order", "config3", args.steps if args.steps <= 500 else 500, 0, None),) + plan
        for key, name, k_steps, w_steps, pol in plan:
            r, e_, _ = measure(name, WORKLOADS[name][5], "rk45", "fused", k_steps, w_steps, 0, False, 0, rank, world, dev, 0.2,
                               args.stream_shards, pol or "trained", args.schedule,
                               want_sustained=(key in ("config2", "config3_natural_order")), want_step_api=(key == "config2" and world == 1),
                               placement="natural" if key == "config3_natural_order" else "regroup")
            subs[key] = slice_of(r, name, WORKLOADS[name][5])
            subs[key]["steps"] = k_steps
            del e_
            if rank == 0 and world == 1 and not args.no_cpu_baseline:
                c = cpu_baseline(name, "rk45", 1.5, pol or "trained", steps=100)
                subs[key]["cpu_baseline"] = {k: c[k] for k in ("value", "unit", "cores", "kind", "sample", "single_thread_rate",
                                                               "fp32_build")}
        if "config2" in subs:
            subs["config2"]["note"] = ("rounds 1-4's headline: `value` = every timed region ONE 20-step launch (the driver's K), "
                                       "`sustained` = 250-step launches")
        detail["configs"] = subs
        out["configs"] = {k: compact(v) for k, v in subs.items()}
    # the two GEMM-shaped rows that are not a BASELINE config of their own: config 4's dense-tail shape
    # (rl_nn_model.py 800 -> 64 -> 64 -> 3 on synthetic U(-1,1) features, 131 072 samples) and one PPO update epoch over a
    # config-5 horizon (65 536 envs x 128 steps = 8.4 M samples: fused-loss backward, weight gradients, Adam, both heads)
    if default_run and world == 1 and not args.no_subconfigs:
        nr = next_rows(dev, args.policy)
        detail["next_rows"] = nr
        out["next_rows"] = {"dense_tail_800_64_64_3": {k: nr["dense_tail_800_64_64_3"][k] for k in ("samples", "kernel_us", "hbm_frac", "mfma_pipe_busy_at_nominal_clock")},
                            "ppo_update_epoch": {k: nr["ppo_update_epoch"][k] for k in ("samples", "seconds", "samples_per_s", "hbm_GB_per_epoch", "hbm_GBps")}}
    if world > 1:
        out["rccl_ranks"] = dist.get_world_size() if dist.get_backend() == "nccl" else 0
        out["dist_backend"] = dist.get_backend()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, args.integrator, args.cpu_seconds, args.policy)
    if rank == 0:
        if detail:
            print(json.dumps({"bench_detail": detail}), flush=True)
        print(json.dumps(out), flush=True)   # the contract line: last
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
