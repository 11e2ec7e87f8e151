"""GPU box: the learner's kernels one by one at config-5 scale (65 536 envs x 128 steps = 8.4 M samples): weight
gradients dW = dZ A^T for the three layer shapes, and the backward pass, each against its fp32-MFMA floor."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from autonomous_quadrotor_environment_amd import _lib as L

N, CH = 65536, 128
dev = "cuda"
big = torch.randn(CH, 128, N, device=dev)
obs = torch.randn(CH, 75, N, device=dev)
small = torch.randn(CH, 4, N, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
peak = 157.3e12
for name, A, m, B, n in (("dW2 128x128", big, 128, big, 128), ("dW1 128x75", big, 128, obs, 75), ("dW3 4x128", small, 4, big, 128)):
    Cm = torch.zeros(m, n, device=dev); rs = torch.zeros(m, device=dev)
    def run():
        L.check(L.lib.qenv_weight_grad(N, CH, m, n, L.ptr(A), L.ptr(B), L.ptr(Cm), L.ptr(rs), stream), "wgrad")
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    mt, nt = (m + 31) // 32, (n + 31) // 32
    floor_ms = (N * CH / 64) * mt * nt * 32 * 64 / 1024 / 2.4e9 * 1e3  # executed 32x32x2 MFMAs / 1024 SIMDs x 64 cycles
    print(json.dumps(dict(kernel=name, ms=round(ms, 3), algorithmic_TFLOPs=round(2 * m * n * N * CH / ms / 1e9, 1),
                          mfma_floor_ms=round(floor_ms, 3), hbm_read_GB=round((m + n) * N * CH * 4 / 1e9, 2),
                          hbm_TBps=round((m + n) * N * CH * 4 / ms / 1e9, 2))), flush=True)
