"""GPU box, diagnostic build (tools/build_variant.sh dt3stamps "-DQENV_DT3_STAMPS" qenv_ppo.hip; QENV_LIB=build_variants/libqenv_dt3stamps.so):
s_memtime stamps of one mid-launch chunk of k_dense_tail_v3 per wave (800 -> 64 -> 64 -> 3, 131 072 samples)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd import _lib as L
g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
K, H, n_out, N = 800, 64, 3, 131072
W = [r(H, K) / K ** 0.5, r(H) * 0.1, r(H, H) / H ** 0.5, r(H) * 0.1, r(n_out, H) / H ** 0.5, r(n_out) * 0.1]
x = torch.rand(K, N, device="cuda", generator=g) * 2 - 1
rows = len(sys.argv) > 1 and sys.argv[1] == "rows"
if rows:
    x = x.t().contiguous()
for _ in range(3):
    aq.dense_tail(x, W, final_tanh=True, rows=rows)
torch.cuda.synchronize()
fn = C.CDLL(L.LIB_PATH).qenv_debug_dt3_stamps
buf = (C.c_ulonglong * 64)()
assert fn(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(8, 8)
names = ["split W1 + features (chunk top -> barrier)", "barrier wait", "issue next chunk's loads", "48 products (+ 24 fragment reads)"]
out = {nm: [int(st[w, k + 1] - st[w, k]) for w in range(8)] for k, nm in enumerate(names)}
out["chunk period (products done -> products done, next chunk)"] = [int(st[w, 5] - st[w, 4]) for w in range(8)]
out["layout"] = "rows [N][K]" if rows else "feature-major [K][N]"
print(json.dumps(out))
