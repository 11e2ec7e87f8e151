"""End-to-end on-device PPO training of the inner-loop controller (environment/controller/ppo.py's loop):
rollout (Philox-sampled Gaussian policy, masked auto-reset) -> GAE -> K_epochs x {loss, backward, Adam}.
Prints one JSON line per update; --eval-every runs deterministic 500-step evaluation episodes like
ppo.py:272-296 (fraction solved = the reference's "Solved Avg" column)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq

p = argparse.ArgumentParser()
p.add_argument("--envs", type=int, default=2048)
p.add_argument("--horizon", type=int, default=64)
p.add_argument("--updates", type=int, default=100)
p.add_argument("--hidden", type=int, default=128)
p.add_argument("--eval-every", type=int, default=10)
p.add_argument("--eval-envs", type=int, default=1024)
p.add_argument("--lr", type=float, default=5e-4)
p.add_argument("--seed", type=int, default=1)
p.add_argument("--integrator", default="rk45")
p.add_argument("--robust", action="store_true", help="train under robust_control: a new perturbed plant per episode + gusts")
p.add_argument("--eval-robust", action="store_true", help="evaluate on the perturbed plant as well (else nominal)")
p.add_argument("--separate-reset", action="store_true", help="auto-reset through sample_init_states + reset(mask) instead of the fused launch")
p.add_argument("--deterministic", action="store_true", help="fixed-order gradient / moment sums: the run is bit-reproducible")
p.add_argument("--digest", action="store_true", help="print a SHA-256 of the final weights (two --deterministic runs must agree)")
p.add_argument("--log", default="")
p.add_argument("--handle", default="f32", choices=["f32", "lean", "f64"], help="env handle (lean = the parity-qualified one)")
p.add_argument("--learn-std", action="store_true", help="train the action std too (model.py fixed_std=False)")
p.add_argument("--inband", action="store_true", help="the in-band warm-up schedule (a restart's warm-up steps ride the next T steps; rows flagged "
                                                      "invalid) with the observations stored as slices: a rollout step is policy + env step")
a = p.parse_args()
if (a.robust or a.eval_robust) and a.handle == "lean":
    p.error("--robust / --eval-robust need --handle f32 or f64: the lean handle has no robust_control kernels")

w = aq.ActorCriticWeights.random(a.hidden, 75, 4, seed=a.seed)
w.std = 0.1
HKW = {"f32": dict(dtype=torch.float32), "lean": dict(dtype=torch.float64, compute="lean"), "f64": dict(dtype=torch.float64)}[a.handle]
env = aq.BatchedQuad(a.envs, 0.01, 1000, training=True, direct_control=1, T=5, integrator=a.integrator, **HKW)
env.robust_control = a.robust  # PolicyRollout then draws the episode parameters / gusts from Philox at every (re)start
ro = aq.PolicyRollout(env, w, a.horizon, store_obs=True, auto_reset=True, seed=a.seed, inband=a.inband)
ro.keep_init = a.separate_reset
learner = aq.PpoLearner(w, lr=a.lr, deterministic=a.deterministic, learn_std=a.learn_std)
ro.start()
log = open(a.log, "w") if a.log else None
t0 = time.time()
total_steps = 0
for u in range(1, a.updates + 1):
    ro.t = 0
    if a.inband:
        ro.run(a.horizon)   # the whole horizon as one library call
    else:
        for _ in range(a.horizon):
            ro.step()
    ret, adv = ro.finish(ddof=0, group=False, bootstrap=True, deterministic=a.deterministic)  # horizon-truncated episodes bootstrap from the critic
    losses = learner.update(ro, ret, adv)
    total_steps += a.envs * a.horizon
    rec = dict(update=u, env_steps=total_steps, wall_s=round(time.time() - t0, 2), mean_reward=float(ro.rewards.mean()),
               episodes=int(env.episodes.sum()), mean_last_return=float(env.last_return.mean()),
               loss_first=float(losses[0]["loss"]), loss_last=float(losses[-1]["loss"]))
    if a.eval_every and (u % a.eval_every == 0 or u == 1):
        ev = aq.BatchedQuad(a.eval_envs, 0.01, 1000, training=False, direct_control=1, T=5, integrator=a.integrator, **HKW)
        ev.robust_control = a.eval_robust
        er = aq.PolicyRollout(ev, w.actor_only(), 500, seed=a.seed + 77)
        er.start(ev.sample_init_states(a.seed + 1000, u))
        er.run_deterministic(500)   # the evaluation episode as one library call (qenv_policy_rollout)
        v = torch.linalg.norm(ev.state[1:6:2], dim=0)
        rec.update(eval_solved=float(ev.solved.float().mean()), eval_v_lt_0p05=float((v < 0.05).float().mean()),
                   eval_broken=float(ev.done.float().mean()), eval_return=float(ev.episode_return.mean()))
    line = json.dumps(rec)
    print(line, flush=True)
    if log:
        log.write(line + "\n"); log.flush()
if a.digest:
    import hashlib
    h = hashlib.sha256()
    for t in w.actor + w.critic:
        h.update(t.detach().cpu().numpy().tobytes())
    print(json.dumps(dict(weights_sha256=h.hexdigest(), deterministic=a.deterministic)))
