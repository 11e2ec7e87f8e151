"""GPU box: the fused rollout launch (k_ppo_fused: policy + env step of a whole horizon in one launch) against the two launches
per step it replaces, at config 4's and config 5's per-GPU sizes: bit-equality of the buffers (this process: step() calls never
use the fused launch) and the time per step of each path (the separate-launch path of run() / run_deterministic() needs
QENV_PPO_FUSED=0, read once per process: the parent starts a child for it).

    python tests/tools/time_fused_rollout.py [envs_config4] [envs_config5] [steps]
"""
import os, subprocess, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch


def weights(aq, dev):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "g9_mlp_N128.npz"))
    w = aq.ActorCriticWeights.from_state_dict({k: g[k] for k in g.files if k.startswith(("actor", "critic"))})
    w.std = 0.1
    return w


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    import autonomous_quadrotor_environment_amd as aq
    n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    n5 = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    dev = torch.device("cuda:0")
    w = weights(aq, dev)
    lean = dict(dtype=torch.float64, compute="lean")
    out = {"fused": os.environ.get("QENV_PPO_FUSED", "1")}
    # config 4: deterministic policy in the loop
    env = aq.BatchedQuad(n4, 0.01, 500, training=True, direct_control=1, T=5, device=dev, **lean)
    ro = aq.PolicyRollout(env, aq.ActorCriticWeights(w.actor, None, dev), K)
    ro.start()
    def c4():
        ro.t = 0
        ro.run_deterministic(K)
    out["config4_us_per_step"] = timed(c4, 5) / K * 1e6
    out["config4_G"] = n4 / out["config4_us_per_step"] * 1e-3
    # config 5: sampling in-band rollout, slices
    env5 = aq.BatchedQuad(n5, 0.01, 500, training=True, direct_control=1, T=5, device=dev, **lean)
    ro5 = aq.PolicyRollout(env5, w, 128, store_obs=True, auto_reset=True, seed=5, inband=True)
    ro5.start()
    def c5():
        ro5.t = 0
        ro5.run(128)
    out["config5_rollout_us_per_step"] = timed(c5, 5) / 128 * 1e6
    def c5s():   # the same horizon as 128 step() calls (qenv_ppo_step: with the fused path each its own launch)
        ro5.t = 0
        for _ in range(128):
            ro5.step()
    out["config5_step_api_us_per_step"] = timed(c5s, 3) / 128 * 1e6
    def c5f():
        ro5.t = 0
        ro5.run(128)
        ro5.finish(ddof=0, group=False)
    out["config5_with_finish_us_per_step"] = timed(c5f, 5) / 128 * 1e6
    out["config5_G"] = n5 / out["config5_with_finish_us_per_step"] * 1e-3
    print(json.dumps(out), flush=True)
    if out["fused"] != "0" and not os.environ.get("NO_CHILD"):
        # bit-equality against step() calls (separate launches) on small batches, ragged sizes included
        for n in (2000, 4160, 64 * 33 + 17):
            res = []
            for mode in ("step", "run"):
                e = aq.BatchedQuad(n, 0.01, 13, training=True, direct_control=1, T=5, device=dev, **lean)
                e.set_ppo_fused(mode != "step")
                r = aq.PolicyRollout(e, w, 40, store_obs=True, auto_reset=True, seed=21, env_offset=500, inband=True)
                r.start()
                if mode == "run":
                    r.run(17); r.run(23)
                else:
                    for _ in range(40):
                        r.step()
                ret, adv = r.finish(ddof=0, group=False, deterministic=True)
                res.append([x.clone() for x in (r.actions, r.logprobs, r.rewards, r.masks, r.obs, r.values, ret, adv, e.state, e.last_return,
                                                e.episodes, e.window, e.i, r.valid)])
            bad = [k for k, (x, y) in enumerate(zip(*res)) if not torch.equal(x, y)]
            print("inband n", n, "mismatching buffers:", bad, flush=True)
            res = []
            for mode in ("step", "run"):
                e = aq.BatchedQuad(n, 0.01, 500, training=True, direct_control=1, T=5, device=dev, **lean)
                e.set_ppo_fused(mode != "step")
                r = aq.PolicyRollout(e, aq.ActorCriticWeights(w.actor, None, dev), 30)
                r.start()
                if mode == "run":
                    r.run_deterministic(11); r.run_deterministic(19)
                else:
                    for _ in range(30):
                        r.step(deterministic=True)
                res.append([x.clone() for x in (r.actions, r.rewards, r.masks, r.done, e.state, e.window, e.i, e.ang, e.ang_vel)])
            bad = [k for k, (x, y) in enumerate(zip(*res)) if not torch.equal(x, y)]
            print("deterministic n", n, "mismatching buffers:", bad, flush=True)
        env_c = dict(os.environ, QENV_PPO_FUSED="0", NO_CHILD="1")
        subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env_c, check=False)


if __name__ == "__main__":
    main()
