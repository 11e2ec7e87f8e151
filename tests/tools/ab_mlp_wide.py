"""GPU box: the N = 256 policy forward: round 2's LDS-staged fp32-MFMA wide kernel (QENV_MLP_WIDE=2) against the kernel it
replaced (QENV_MLP_WIDE=0), bit for bit, and the launch times of both next to round 3's split-bf16 kernel (the default, T = 5
only; its own accuracy check against fp64 torch is tests/tools/fuzz_mlp.py).  Each kernel runs in its own process (the switch
is read once).  python tests/tools/ab_mlp_wide.py"""
import json, os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    import autonomous_quadrotor_environment_amd as aq
    res = {}
    torch.manual_seed(11)
    for T in (5, 3, 8):
        w = aq.ActorCriticWeights.random(256, 15 * T, 4, device="cuda", seed=3)
        for N in (1, 100, 4096, 131072):
            env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=T, dtype=torch.float32)
            env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
            for k in range(T + 1):
                env.step(torch.rand(4, N, device="cuda") * 2 - 1, push_window=True)
            out = torch.empty(4, N, device="cuda")
            env.mlp_forward(w, "actor", out=out)
            val = torch.empty(1, N, device="cuda")
            env.mlp_forward(w, "critic", out=val)
            logp = torch.empty(4, N, device="cuda")
            act = env.policy_act(w, seed=5, step=2, logprob=logp)
            torch.cuda.synchronize()
            h = hashlib.sha256(out.cpu().numpy().tobytes() + val.cpu().numpy().tobytes() + act.cpu().numpy().tobytes()
                               + logp.cpu().numpy().tobytes()).hexdigest()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50 if N > 4096 else 200
            for _ in range(5):
                env.mlp_forward(w, "actor", out=out)
            a.record()
            for _ in range(reps):
                env.mlp_forward(w, "actor", out=out)
            b.record(); torch.cuda.synchronize()
            res[f"T{T}_N{N}"] = dict(sha=h[:16], us=round(a.elapsed_time(b) * 1000 / reps, 2), finite=bool(torch.isfinite(out).all()))
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
        sys.exit(0)
    runs = {}
    for wide in ("1", "2", "0"):
        env = dict(os.environ, QENV_MLP_WIDE=wide)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        if p.returncode != 0:
            print(p.stdout[-2000:], p.stderr[-4000:]); sys.exit(1)
        runs[wide] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    bad = 0
    for k in runs["2"]:
        w, o, v3 = runs["2"][k], runs["0"][k], runs["1"][k]
        same = w["sha"] == o["sha"] and w["finite"]
        bad += (not same) or (not v3["finite"])
        print(f"{k:12s} default {v3['us']:9.2f} us   wide(fp32) {w['us']:9.2f} us   old {o['us']:9.2f} us   wide == old bit for bit: {same}")
    sys.exit(1 if bad else 0)
