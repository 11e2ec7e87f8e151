"""GPU box: launch time of one full env step (scipy-exact RK45, actions from a buffer, the reference's reward/termination,
window push) at one batch size, on live closed-loop states (the trained actor drives the envs between timed blocks so the
adaptive integrator sees config 4's attempt counts).  QENV_STEP_GRID_LANES / QENV_LIB select A/B variants.
python tests/tools/time_step_one.py [envs] [dtype=f64|f32]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "g9_mlp_N128.npz"))
kw = dict(dtype=torch.float32) if f32 else {}
env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5, **kw)
env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
act = (torch.rand(4, N, device="cuda", dtype=env.dtype) * 2 - 1) * 0.3
for _ in range(30):
    env.step(act, push_window=True)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        env.step(act, push_window=True)
    b.record(); torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) * 1000 / 20)
print(json.dumps(dict(lib=os.environ.get("QENV_LIB", "default"), lanes=os.environ.get("QENV_STEP_GRID_LANES", "auto"),
                      envs=N, dtype="f32" if f32 else "lean", us=round(best, 2))))
