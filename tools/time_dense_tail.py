"""GPU box: the dense tail of visual_landing/rl_nn_model.py (800 -> 64 -> 64 -> 3, tanh) on synthetic U(-1,1) features
at config 4's per-GPU batch (the GEMM-shape run): time, HBM rate (the 800 input floats per sample are the
traffic), algorithmic TFLOP/s."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import autonomous_quadrotor_environment_amd as aq

g = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, device="cuda", generator=g)
SHAPES = ((800, 64, 3, 131072), (800, 64, 3, 1048576), (800, 128, 3, 131072))
for K, H, n_out, N in SHAPES[:int(sys.argv[1]) if len(sys.argv) > 1 else 3]:   # (argv[1] = 1: config 4's shape only -- profiling runs)
    W = [r(H, K) / K ** 0.5, r(H) * 0.1, r(H, H) / H ** 0.5, r(H) * 0.1, r(n_out, H) / H ** 0.5, r(n_out) * 0.1]
    x = torch.rand(K, N, device="cuda", generator=g) * 2 - 1
    out = torch.empty(n_out, N, device="cuda")
    for _ in range(3):
        aq.dense_tail(x, W, final_tanh=True, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        aq.dense_tail(x, W, final_tanh=True, out=out)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1000 / 20
    flops = 2.0 * N * (K * H + H * H + H * n_out)
    rec = dict(K=K, hidden=H, n_out=n_out, samples=N, us=round(us, 1), samples_per_s=round(N / us * 1e6),
               hbm_TBps=round((K + n_out) * 4 * N / us / 1e6, 2), algorithmic_TFLOPs=round(flops / us / 1e6, 1))
    if H == 64 and K * N < (1 << 29):   # the same features sample-major ([N, K]: the reference's own layout)
        xr = x.t().contiguous(); out2 = torch.empty_like(out)
        for _ in range(3):
            aq.dense_tail(xr, W, final_tanh=True, out=out2, rows=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20):
            aq.dense_tail(xr, W, final_tanh=True, out=out2, rows=True)
        b.record(); torch.cuda.synchronize()
        us2 = a.elapsed_time(b) * 1000 / 20
        rec.update(rows_us=round(us2, 1), rows_hbm_TBps=round((K + n_out) * 4 * N / us2 / 1e6, 2), rows_bit_identical=bool(torch.equal(out, out2)))
    print(json.dumps(rec), flush=True)
