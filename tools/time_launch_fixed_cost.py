"""GPU box: fixed cost of a fused rollout launch -- duration of k_rollout (HIP events around trains of back-to-back
launches) against the number of env steps per launch, per integrator: time = a + b n.  Lean handle, config 2's size."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
handle = sys.argv[2] if len(sys.argv) > 2 else "lean"
for integ in ("rk45", "dp5", "rk4"):
    env = bench.build_env("config2", handle, integ, n, 0, n, dev)
    pts = {}
    for steps in (1, 2, 5, 10, 20, 50, 100, 250):
        run = lambda: env.rollout("const", steps, flags=1)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        reps = max(20, int(4000 / (4 + 3.4 * steps)))
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                run()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / reps)
        pts[steps] = best
    xs, ys = np.array(list(pts)), np.array(list(pts.values()))
    b_, a_ = np.polyfit(xs[2:], ys[2:], 1)
    print(json.dumps(dict(handle=handle, integrator=integ, envs=n, us_per_launch={k: round(v, 2) for k, v in pts.items()},
                          fixed_us=round(float(a_), 2), us_per_step=round(float(b_), 3))), flush=True)
