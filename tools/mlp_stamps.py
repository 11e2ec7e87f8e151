"""GPU box: where a k_mlp launch spends its time, from in-kernel s_memtime stamps (diagnostic build:
tools/build_variant.sh stamps "-DQENV_MLP_STAMPS"; run with QENV_LIB=build_variants/libqenv_stamps.so).

    QENV_LIB=build_variants/libqenv_stamps.so python tools/mlp_stamps.py [envs]

Per wave: entry, end of the weight staging (barrier), then per 32-env tile: start, window loaded + split, layer 1 done,
layer-1 activations split, layers 2 + 3 done, outputs stored.  Prints the median / p10 / p90 over the waves of every
phase in shader cycles, the clock (cycles per s_memrealtime tick x 100 MHz) and the launch time by HIP events."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

N = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g9_mlp_N128.npz"))
w = aq.ActorCriticWeights.from_state_dict({k: g[k] for k in g.files if k.startswith(("actor", "critic"))})
env = aq.BatchedQuad(N, 0.01, 500, training=True, direct_control=1, T=5)
env.reset(env.sample_init_states(0, 0, 0), return_history=False, push_window=True)
out = torch.empty(4, N, device="cuda", dtype=env.dtype)
for _ in range(20):
    env.mlp_forward(w, "actor", out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100):
    env.mlp_forward(w, "actor", out=out)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) * 10.0
fn = L.lib.qenv_debug_mlp_stamps
fn.restype = C.c_int
n_waves = 256 * 8
buf = (C.c_ulonglong * (n_waves * 16))()
assert fn(buf, n_waves * 16) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(n_waves, 16).astype(np.int64)
s = s[s[:, 0] > 0]
clock_ghz = np.median((s[:, 7 + 6] - s[:, 0]) / np.maximum(1, (s[:, 15] - s[:, 14])) * 0.1)
names = ["staging (entry -> barrier)", "barrier -> tile 1 start", "t1 window load + split", "t1 layer 1 (+tanh)", "t1 split activations",
         "t1 layers 2+3 (+tanh)", "t1 store", "t1 end -> t2 start", "t2 window load + split", "t2 layer 1 (+tanh)", "t2 split activations",
         "t2 layers 2+3 (+tanh)", "t2 store"]
rows = {}
for k, nm in enumerate(names):
    d = (s[:, k + 1] - s[:, k]).astype(np.float64)
    rows[nm] = [float(np.percentile(d, q)) for q in (10, 50, 90)]
# per wave half (waves 0-3 / 4-7 of a workgroup share the SIMDs pairwise): offsets from the wave's entry, medians
wave_in_block = np.arange(n_waves)[: len(s)] % 8 if len(s) == n_waves else None
halves = {}
if wave_in_block is not None:
    for nm, sel in (("lower_waves_0_3", wave_in_block < 4), ("upper_waves_4_7", wave_in_block >= 4)):
        rel = (s[sel][:, 1:14] - s[sel][:, :1]).astype(np.float64)
        halves[nm] = {f"slot{k + 1}": float(np.median(rel[:, k])) for k in range(13)}
total = (s[:, 13] - s[:, 0]).astype(np.float64)
start_spread = float(s[:, 0].max() - s[:, 0].min())
print(json.dumps(dict(lib=os.environ.get("QENV_LIB", "default"), envs=N, launch_us=round(us, 2), clock_ghz_est=round(float(clock_ghz), 3),
                      waves=int(len(s)), wave_total_cycles_p10_50_90=[float(np.percentile(total, q)) for q in (10, 50, 90)],
                      first_to_last_entry_cycles=start_spread, last_exit_minus_first_entry=float(s[:, 13].max() - s[:, 0].min()),
                      phases_cycles_p10_50_90=rows, medians_from_entry_by_wave_half=halves), indent=1))
