"""GPU box: where the headline launch (config 2: 65 536 envs, lean fp64, scipy-exact RK45, constant action, K steps per launch)
spends its time, from in-kernel stamps (diagnostic build: tools/build_variant.sh rstamps "-DQENV_ROLLOUT_STAMPS" qenv_env_lean.hip;
QENV_LIB=build_variants/libqenv_rstamps.so python tests/tools/rollout_stamps.py [K]).  Per wave: entry (wall clock), state rows
loaded, end of steps 1 / 2 / K, rows stored, exit (wall clock); and the host's view of the same launch (HIP events, a
synchronised region)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
env = bench.build_env("config2", "lean", "rk45", 65536, 0, 65536, dev)
run = lambda: env.rollout("const", K, flags=1)
for _ in range(50):
    run()
torch.cuda.synchronize()
regs = []
for _ in range(300):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); regs.append(time.perf_counter() - t0)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200):
    run()
b.record(); torch.cuda.synchronize()
back_to_back_us = a.elapsed_time(b) * 1e3 / 200
torch.cuda.synchronize(); run(); torch.cuda.synchronize()
fn = L.lib.qenv_debug_rollout_stamps
fn.restype = C.c_int
buf = (C.c_ulonglong * (1024 * 8))()
assert fn(buf, 1024 * 8) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
s = s[s[:, 1] > 0]
clk = np.median((s[:, 6] - s[:, 1]) / np.maximum(1, s[:, 7] - s[:, 0])) * 0.1   # shader cycles per 10 ns tick -> GHz
us = lambda c: c / (clk * 1e3)
q = lambda x: [round(float(np.percentile(x, p)), 2) for p in (10, 50, 90, 100)]
out = dict(K=K, waves=len(s), clock_ghz=round(float(clk), 3),
           region_us_synchronised_median=round(float(np.median(regs)) * 1e6, 2), launch_us_back_to_back=round(back_to_back_us, 2),
           entry_spread_us_p10_p50_p90_max=q((s[:, 0] - s[:, 0].min()) * 0.01),
           exit_spread_us_p10_p50_p90_max=q((s[:, 7] - s[:, 0].min()) * 0.01),
           wave_life_us=q((s[:, 7] - s[:, 0]) * 0.01),
           load_us=q(us(s[:, 2] - s[:, 1])), step1_us=q(us(s[:, 3] - s[:, 2])), step2_us=q(us(s[:, 4] - s[:, 3])),
           steps3_to_K_us_per_step=q(us(s[:, 5] - s[:, 4]) / max(1, K - 2)), store_us=q(us(s[:, 6] - s[:, 5])))
print(json.dumps(out, indent=1))
