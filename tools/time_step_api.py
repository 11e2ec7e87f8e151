"""GPU box: host-side cost of the per-step Python API (BatchedQuad.step) against the kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
for N in (1024, 65536):
    env = aq.BatchedQuad(N, 0.01, 10 ** 6, training=False, direct_control=1, T=1)
    env.reset(aq.random_init_states(1, N).T.copy() * 0.1 + np.eye(13)[6][:, None] * 0.9)
    act = torch.zeros(4, N, device="cuda")
    outs = (torch.empty(14, N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda", dtype=torch.uint8))
    for name, fn in (("step(act) fresh outputs", lambda: env.step(act)), ("step(act, out=preallocated)", lambda: env.step(act, out=outs))):
        for _ in range(200): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2000): fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize(); t_all = time.perf_counter() - t0
        print(f"N={N:6d} {name:30s} host issue {t_issue / 2000 * 1e6:6.1f} us/step, wall {t_all / 2000 * 1e6:6.1f} us/step")
