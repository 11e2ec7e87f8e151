"""GPU box: launches the calibration copy kernel (same access shape as the env kernels, known byte
count, footprint far beyond the 256 MiB Infinity Cache) for the rocprofv3 PMC passes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autonomous_quadrotor_environment_amd import _lib as L
N, rows = 16 * 1024 * 1024, 13
src = torch.rand(rows, N, device="cuda", dtype=torch.float32)
dst = torch.empty_like(src)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    L.check(L.lib.qenv_debug_copy_rows(L.F32, L.ptr(dst), L.ptr(src), rows, N, stream), "copy")
torch.cuda.synchronize()
print("calib bytes per launch (read = write):", rows * N * 4)
