"""CPU, world_size 2, gloo: the N>1 path's host logic -- env-range sharding and the all-gather
of per-env episode returns (even and ragged shards) -- plus the 3-scalar moment all-reduce that
keeps GAE's advantage normalisation identical to a single learner."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import ROOT


def _free_port():
    """A rendezvous port the OS hands out (no fixed numbers: two test runs on one host must not collide)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_global, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autonomous_quadrotor_environment_amd.distributed import all_gather_returns, shard_range
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    lo, hi = shard_range(n_global, rank, world)
    # a per-env quantity that is a pure function of the global env index and the seeded init state
    x0 = fast_random_init_states(9, n_global, lo, hi)
    local = torch.from_numpy(x0[:, 1] + np.arange(lo, hi))
    full = all_gather_returns(local, n_global)
    # global advantage moments: sum of per-shard {count, sum, sumsq}
    stats = torch.tensor([hi - lo, float(local.sum()), float((local ** 2).sum())], dtype=torch.float64)
    dist.all_reduce(stats)
    q.put((rank, full.numpy(), stats.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [4096, 1001])
def test_all_gather_returns_world2(n_global):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
    want = fast_random_init_states(9, n_global)[:, 1] + np.arange(n_global)
    for rank, full, stats in res:
        assert np.array_equal(full, want), rank
        assert stats[0] == n_global and abs(stats[1] - want.sum()) < 1e-6 and abs(stats[2] - (want ** 2).sum()) < 1e-3


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autonomous_quadrotor_environment_amd.distributed import GradBucket
    shapes = [(128, 75), (128,), (128, 128), (128,), (4, 128), (4,), (128, 75), (128,), (128, 128), (128,), (1, 128), (1,)]
    params = [torch.zeros(s) for s in shapes]
    b = GradBucket(params)
    for k, v in enumerate(b.views):  # the per-tensor gradients of this rank's shard
        v.copy_(torch.full(v.shape, float(k + 1) * (rank + 1)))
    b.all_reduce_mean_()
    q.put((rank, [float(v.mean()) for v in b.views], int(b.flat.numel()), [tuple(v.shape) for v in b.views] == shapes))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_bucket_all_reduce_world2():
    """The data-parallel learner's one exchange: all 12 gradient tensors travel as ONE 52 869-float bucket and come
    back as the mean over the ranks, in place, through the per-tensor views the Adam kernel reads."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, means, numel, shapes_ok in res:
        assert shapes_ok and numel == 2 * (128 * 75 + 128 + 128 * 128 + 128) + 4 * 128 + 4 + 128 + 1
        assert means == [1.5 * (k + 1) for k in range(12)], rank


def _masked_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from autonomous_quadrotor_environment_amd.ppo import ActorCriticWeights, PpoLearner
    rng = np.random.RandomState(0)
    mk = lambda o: [rng.randn(16, 75), rng.randn(16), rng.randn(16, 16), rng.randn(16), rng.randn(o, 16), rng.randn(o)]
    learner = PpoLearner(ActorCriticWeights(mk(4), mk(1), device="cpu"), K_epochs=1, group=None)
    # per-sample "gradients" of the whole job; the ranks own unequal numbers of VALID samples (in-band masks differ)
    g = torch.from_numpy(np.random.RandomState(1).randn(1000, learner.bucket.flat.numel())).float()
    valid = torch.from_numpy(np.random.RandomState(2).rand(1000) < 0.8)
    valid[:500] &= torch.from_numpy(np.random.RandomState(3).rand(500) < 0.5)   # rank 0's half holds far fewer
    lo, hi = (0, 500) if rank == 0 else (500, 1000)
    n_local = int(valid[lo:hi].sum())
    n_global = learner._global_count(n_local, "cpu")
    learner.bucket.flat.copy_(g[lo:hi][valid[lo:hi]].double().sum(0).float() / n_global)   # what the masked backward leaves
    learner.bucket.all_reduce_sum_(learner.group)
    want = (g[valid].double().sum(0) / int(valid.sum())).float()
    q.put((rank, learner._world(), n_local, n_global, int(valid.sum()), float((learner.bucket.flat - want).abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_masked_data_parallel_gradients_are_the_single_learners_world2():
    """An in-band rollout's valid masks give the ranks DIFFERENT sample counts: the learner all-reduces the count, each
    rank normalises by the global one and the bucket is summed -- the single learner's sum / count (a mean of per-rank
    means would weight rank 0's few samples as much as rank 1's many)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_masked_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    counts = {r[0]: r[2] for r in res}
    assert counts[0] < 0.7 * counts[1]
    for rank, world, n_local, n_global, n_true, err in res:
        assert world == 2 and n_global == n_true == counts[0] + counts[1] and err < 1e-6, (rank, err)
