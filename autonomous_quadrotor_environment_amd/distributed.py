"""Multi-GPU: one process per GPU, env index ranges sharded with no data-path collective.

Envs are fully independent, so rank r of W owns the contiguous global env
range shard_range(N, r, W); initial states are a pure function of (seed, global env index)
(quadrotor_env.fast_random_init_states) so results are shard-invariant.  The only exchange
the path has is the RCCL all-gather of per-env episode returns (and the 3-scalar all-reduce
inside ppo.gae for the global advantage normalisation).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .quadrotor_env import BatchedQuad, fast_random_init_states


def shard_range(n_global, rank, world):
    """Contiguous [lo, hi) of the global env index range owned by `rank`."""
    base, rem = divmod(int(n_global), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_process_group_from_env(backend=None):
    """torch.distributed init from RANK / WORLD_SIZE / MASTER_* (torchrun).  backend 'nccl' is
    RCCL on ROCm.  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())  # rehearsals with more ranks than GPUs (gloo only)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # every launcher (torchrun, bench.py's self-launch, the tests) sets it: a
            raise RuntimeError(              # silently assumed fixed port would let two jobs on one host meet
                "WORLD_SIZE > 1 but MASTER_PORT is not set: launch the ranks with `python -m torch.distributed.run "
                "--master-addr 127.0.0.1 --master-port <free port> ...` (or `bench.py --gpus N`, which picks one)")
        if backend is None:
            backend = os.environ.get("QENV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def all_gather_returns(local_returns, n_global=None, group=None):
    """All-gather of per-env episode returns: every rank ends with the full [n_global] vector
    in global env order (ragged shards are padded to the largest one on the wire)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_returns.clone()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_local = local_returns.numel()
    if n_global is None:
        cnt = torch.tensor([n_local], device=local_returns.device, dtype=torch.int64)
        dist.all_reduce(cnt, group=group)
        n_global = int(cnt.item())
    sizes = [shard_range(n_global, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    if all(hi - lo == max_n for lo, hi in sizes):
        out = torch.empty(n_global, device=local_returns.device, dtype=local_returns.dtype)
        dist.all_gather_into_tensor(out, local_returns.contiguous(), group=group)
        return out
    pad = torch.zeros(max_n, device=local_returns.device, dtype=local_returns.dtype)
    pad[:n_local] = local_returns
    buf = torch.empty(world * max_n, device=local_returns.device, dtype=local_returns.dtype)
    dist.all_gather_into_tensor(buf, pad, group=group)
    return torch.cat([buf[r * max_n:r * max_n + (hi - lo)] for r, (lo, hi) in enumerate(sizes)])


class GradBucket:
    """One flat fp32 buffer holding every parameter gradient of the actor and the critic (53 k floats = 212 KB for
    N=128): the data-parallel learner averages it with a single all-reduce (RCCL over xGMI: latency-bound at this
    size, so one bucket, not one collective per tensor).  `views[i]` aliases the slice of parameter i."""

    def __init__(self, params):
        total = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def all_reduce_mean_(self, group=None):
        """Average over the ranks (each rank's loss is a mean over its equal-sized shard of the batch).  No-op
        without an initialised process group or with one rank."""
        if group is False or not (dist.is_available() and dist.is_initialized()):
            return self.flat
        world = dist.get_world_size(group)
        if world > 1:
            dist.all_reduce(self.flat, group=group)
            self.flat.div_(world)
        return self.flat

    def all_reduce_sum_(self, group=None):
        """Sum over the ranks: for gradients that every rank already normalised by the GLOBAL sample count (shards with
        different numbers of valid rows -- an in-band rollout's masks).  No-op without a process group or with one rank."""
        if group is False or not (dist.is_available() and dist.is_initialized()):
            return self.flat
        if dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, group=group)
        return self.flat


class ShardedQuad:
    """This rank's slice of a global batch of `n_global` envs."""

    def __init__(self, n_global, *args, rank=None, world=None, device=None, **kw):
        self.rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        self.world = dist.get_world_size() if world is None and dist.is_initialized() else (world or 1)
        self.n_global = int(n_global)
        self.lo, self.hi = shard_range(n_global, self.rank, self.world)
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self.env = BatchedQuad(self.hi - self.lo, *args, device=device, **kw)

    def reset_seeded(self, seed):
        x0 = fast_random_init_states(seed, self.n_global, self.lo, self.hi)
        return self.env.reset(torch.from_numpy(np.ascontiguousarray(x0.T)))

    def gather_returns(self, group=None):
        return all_gather_returns(self.env.episode_return, self.n_global, group)
