"""Developer diagnostic: per-component error of reset(det)+step vs golden G4 (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from tests.conftest import golden
np.set_printoptions(linewidth=200)
g = golden("g4_step_direct")
n = len(g["x0"])
for dt in (torch.float32, torch.float64):
    for integ in ("rk45", "dp5"):
        env = aq.BatchedQuad(n, 0.01, 1000, T=1, dtype=dt, direct_control=1, integrator=integ)
        st, _ = env.reset(torch.from_numpy(g["x0"].T.copy()))
        got = st[0].double().cpu().numpy().T
        ref = g["r_quat_state"]
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
        print(dt, integ, "per-component max rel err:", np.array2string(err.max(0), precision=1))
        srt = np.sort(err.max(1))
        print("   sample-wise max err percentiles 50/90/99/100:", srt[n // 2], srt[int(n * .9)], srt[int(n * .99)], srt[-1])
        k = np.unravel_index(err.argmax(), err.shape)
        print("   worst sample", k, "got", got[k], "ref", ref[k], "nfev", g["nfev_reset"][k[0]], "x0", g["x0"][k[0]])
