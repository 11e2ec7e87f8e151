import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import autonomous_quadrotor_environment_amd as aq
N, S = 65536, 128
v = torch.randn(S + 1, N, device="cuda"); m = (torch.rand(S, N, device="cuda") > 0.01).float(); r = torch.randn(S, N, device="cuda")
for _ in range(3): aq.gae(v, m, r, ddof=0, group=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): aq.gae(v, m, r, ddof=0, group=False)
e1.record(); e1.synchronize()
print("gae + normalise us:", e0.elapsed_time(e1) / 20 * 1e3)
