"""GPU box: the staged policy kernel (k_mlp_v3) against the tile-per-workgroup one (k_mlp_small) on the same observations,
ring-free (qenv_dense_tail_chunks), for several batch sizes / chunk counts / heads.  QENV_MLP_SMALL_TILES is read once per
process, so each kernel runs in a child process.  python tests/tools/ab_mlp_kernels.py"""
import json, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [(500, 1, 1), (500, 24, 1), (500, 24, 4), (512, 24, 1), (1000, 24, 1), (1000, 24, 4), (333, 24, 1), (4096, 3, 1)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from autonomous_quadrotor_environment_amd import _lib as L
    import ctypes as C
    out = {}
    for (N, nc, n_out) in CASES:
        g = torch.Generator(device="cuda").manual_seed(N + nc + n_out)
        r = lambda *s: torch.randn(*s, device="cuda", generator=g)
        W = [r(128, 75) * 0.2, r(128) * 0.1, r(128, 128) * 0.15, r(128) * 0.1, r(n_out, 128) * 0.2, r(n_out) * 0.1]
        x = r(nc, 75, N).contiguous()
        y = torch.empty(nc, n_out, N, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(L.lib.qenv_dense_tail_chunks(N, nc, 75, 128, n_out, int(n_out == 4), L.ptr(x), *[L.ptr(t) for t in W], L.ptr(y), st), "chunks")
        torch.cuda.synchronize()
        np.save(os.path.join(sys.argv[2], f"y_{N}_{nc}_{n_out}.npy"), y.cpu().numpy())
    sys.exit(0)

import tempfile
res = {}
dirs = {}
for tag, tiles in (("staged", "0"), ("small", "100000000")):
    d = tempfile.mkdtemp()
    dirs[tag] = d
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", d], env=dict(os.environ, QENV_MLP_SMALL_TILES=tiles),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
for (N, nc, n_out) in CASES:
    a = np.load(os.path.join(dirs["staged"], f"y_{N}_{nc}_{n_out}.npy")); b = np.load(os.path.join(dirs["small"], f"y_{N}_{nc}_{n_out}.npy"))
    bad = np.argwhere(a != b)
    res[f"N{N}_chunks{nc}_out{n_out}"] = dict(mismatches=int(len(bad)), max_abs=float(np.abs(a - b).max()),
                                              first=[int(v) for v in bad[0]] if len(bad) else None,
                                              chunks_hit=sorted(set(int(v[0]) for v in bad))[:8], envs_hit=sorted(set(int(v[2]) for v in bad))[:12])
print(json.dumps(res, indent=1))
