"""GPU box: where a tile of k_mlp_bwd_v3 spends its time, from in-kernel s_memtime stamps of each wave's third tile (diagnostic
build: tools/build_variant.sh bwdstamps "-DQENV_BWD_STAMPS" qenv_train.hip; QENV_LIB=build_variants/libqenv_bwdstamps.so)."""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

N, S = 65536, 32
w = aq.ActorCriticWeights.random(128, 75, 4, seed=1); w.std = 0.1
env = aq.BatchedQuad(N, 0.01, 1000, training=True, direct_control=1, T=5, dtype=torch.float32)
ro = aq.PolicyRollout(env, w, S, store_obs=True, auto_reset=True, seed=3)
ro.start()
for _ in range(S):
    ro.step()
ret, adv = ro.finish(ddof=0, group=False, bootstrap=True)
learner = aq.PpoLearner(w, K_epochs=1)
learner.update(ro, ret, adv); learner.update(ro, ret, adv)
torch.cuda.synchronize()
fn = L.lib.qenv_debug_bwd_stamps
fn.restype = C.c_int
buf = (C.c_ulonglong * (1024 * 16))()
assert fn(buf, 1024 * 16) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16).astype(np.int64)
s = s[s[:, 0] > 0]
names = ["observation split (+ next tile's request)", "layer 1 (3 pipelined chains + stores of H1)", "layer 2 + layer 3",
         "output, loss, dz3", "dZ2 of step 0", "W2^T dZ2 walk (8 steps, dZ2 / H2 stores inside)", "dZ1 + stores"]
out = {"waves": int(len(s)), "tile_cycles_p10_50_90": [float(np.percentile(s[:, 7] - s[:, 0], q)) for q in (10, 50, 90)]}
for k, nm in enumerate(names):
    d = (s[:, k + 1] - s[:, k]).astype(np.float64)
    out[nm] = [float(np.percentile(d, q)) for q in (10, 50, 90)]
print(json.dumps(out, indent=1))
