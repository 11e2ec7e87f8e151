"""GPU box: wall time per step of the N = 1 drop-in `quad` (one launch + one small D2H copy per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import autonomous_quadrotor_environment_amd as aq
env = aq.quad(0.01, 5000, training=True, euler=0, direct_control=0, T=5, clipped=True); env.seed(1)
env.reset(); ctl = aq.pid_control(env); a = np.array([9.82 * 1.03, 0, 0, 0])
for _ in range(50):
    env.step(a); a = ctl.control(np.zeros(3), np.zeros(3), 0, 0)
t0 = time.perf_counter()
for _ in range(2000):
    env.step(a); a = ctl.control(np.zeros(3), np.zeros(3), 0, 0)
dt = time.perf_counter() - t0
print(f"quad drop-in with the PID controller: {dt / 2000 * 1e6:.0f} us per step = {2000 / dt:.0f} steps/s (reference: ~614 steps/s on this loop)")
