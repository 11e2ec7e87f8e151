"""GPU box, diagnostic build (tools/build_variant.sh g5stamps "-DQENV_GRAD5_STAMPS" qenv_train.hip; QENV_LIB=build_variants/libqenv_g5stamps.so):
s_memtime stamps of one steady-state tile of k_mlp_grad_v5 per wave and phase, at config 5's horizon."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L

N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = aq.ActorCriticWeights.random(128, 75, 4, seed=1); w.std = 0.1
g = torch.Generator(device="cuda").manual_seed(0)
obs = torch.randn(S, 75, N, device="cuda", generator=g) * 0.5
act = torch.randn(S, 4, N, device="cuda", generator=g) * 0.3
lp = torch.randn(S, 4, N, device="cuda", generator=g) * 0.1 + 1.3
adv, ret = torch.randn(S, N, device="cuda", generator=g), torch.randn(S, N, device="cuda", generator=g)
ln = aq.PpoLearner(w, K_epochs=1, group=False)
for _ in range(3):
    ln.gradients(obs, act, lp, adv, ret)
torch.cuda.synchronize()
fn = L.lib._lib.qenv_debug_grad5_stamps if hasattr(L.lib, "_lib") else C.CDLL(L.LIB_PATH).qenv_debug_grad5_stamps
buf = (C.c_ulonglong * 128)()
assert fn(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(8, 16)
names = ["L1 chain", "tanh+split+H1 writes", "B1 wait", "L2 chain", "tanh+l3+H2 writes", "B2 wait", "loss+dz3+dZ2 writes", "B3 wait",
         "g3 (dW3)", "W2T dZ2 chain", "dZ1+dW1", "Bx wait", "store_x+dW2+db2", "B0 wait"]
out = {"cycles_per_tile_per_wave": [int(st[w_, 14] - st[w_, 0]) for w_ in range(8)]}
for k, nm in enumerate(names):
    out[nm] = [int(st[w_, k + 1] - st[w_, k]) for w_ in range(8)]
print(json.dumps(out))
