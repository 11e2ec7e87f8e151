"""GPU box, diagnostic build (tools/build_variant.sh rstamps "-DQENV_ROLLOUT_STAMPS" qenv_env_lean.hip;
QENV_LIB=build_variants/libqenv_rstamps.so): the phases of ONE env step in the middle of a k_rollout launch, per wave --
controller law + mixing | integrator | Euler angles | done / reward | epilogue -- for config 3 (LQR, 262 144 envs, 250-step
launches from the random initial states) or config 2 (constant action, 65 536 envs).

    python tests/tools/rollout_phase_stamps.py config3|config2 [K]
"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from autonomous_quadrotor_environment_amd import _lib as L

wl = sys.argv[1] if len(sys.argv) > 1 else "config3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 250
n, _, ctrl, flags, _, _, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
env = bench.build_env(wl, "lean", "rk45", n, 0, n, dev)
x0 = torch.from_numpy(np.ascontiguousarray(bench.make_init(wl, n, 0, n).T)).to(dev)
for _ in range(3):
    env.reset(x0)
    env.rollout(ctrl, K, flags=flags)
torch.cuda.synchronize()
fn = L.lib.qenv_debug_rollout_phase
fn.restype = C.c_int
buf = (C.c_ulonglong * (1024 * 8))()
assert fn(buf, 1024 * 8) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
s = s[s[:, 0] > 0]
names = ["controller law + mixing", "integrator", "Euler angles", "done / reward", "epilogue (window push)"]
q = lambda x: [int(np.percentile(x, p)) for p in (10, 50, 90)]
out = {"workload": wl, "steps_per_launch": K, "stamped_step": K // 2, "waves": len(s),
       "cycles_p10_p50_p90": {nm: q(s[:, k + 1] - s[:, k]) for k, nm in enumerate(names)},
       "whole_step": q(s[:, 5] - s[:, 0])}
print(json.dumps(out, indent=1))
