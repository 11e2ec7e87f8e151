"""GPU box, 2 ranks on the one GPU over gloo (`QENV_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2
--master-addr 127.0.0.1 --master-port 29533 tools/dp_learner_rehearsal.py`): the data-parallel learner.  Each rank rolls
out its half of the env range (shard-invariant Philox streams), GAE normalises with all-reduced moments, gradients are
averaged with one bucketed all-reduce per epoch.  Checks: (i) both ranks end with bit-identical weights, (ii) they match
a single-process run over the whole env range to summation-order noise."""
import os, sys, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd.distributed import init_process_group_from_env, shard_range

rank, world, local = init_process_group_from_env()
N, H, U = 4096, 16, 3
# --inband: the in-band warm-up schedule (valid masks: the ranks hold DIFFERENT numbers of valid rows, the learner normalises
# by the all-reduced count and sums the gradients); short episodes so that many envs restart inside the horizon
INBAND = "--inband" in sys.argv


def run(lo, hi, group):
    w = aq.ActorCriticWeights.random(64, 75, 4, seed=5)
    env = aq.BatchedQuad(hi - lo, 0.01, 12 if INBAND else 60, training=True, direct_control=1, T=5, dtype=torch.float32)
    ro = aq.PolicyRollout(env, w, H, store_obs=True, auto_reset=True, seed=3, env_offset=lo, inband=INBAND)
    ln = aq.PpoLearner(w, K_epochs=2, group=group, deterministic=True)
    ro.start()
    for _ in range(U):
        ro.t = 0
        for _ in range(H):
            ro.step()
        ret, adv = ro.finish(bootstrap=True, group=group, deterministic=True)
        ln.update(ro, ret, adv)
    return torch.cat([t.reshape(-1) for t in w.actor + w.critic])

lo, hi = shard_range(N, rank, world)
w_dp = run(lo, hi, None if world > 1 else False)
digest = hashlib.sha256(w_dp.cpu().numpy().tobytes()).hexdigest()
out = dict(rank=rank, world=world, shard=[lo, hi], weights_sha256=digest[:16], inband=INBAND)
if world > 1:
    digs = [None] * world
    dist.all_gather_object(digs, digest)
    out["ranks_identical"] = len(set(digs)) == 1
    w_full = run(0, N, False)  # the same job on one learner
    out["max_abs_diff_vs_single_learner"] = float((w_dp - w_full).abs().max())
    out["max_abs_weight"] = float(w_full.abs().max())
print(json.dumps(out), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
