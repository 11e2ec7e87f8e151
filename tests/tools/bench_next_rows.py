"""GPU box: throughput of the SURVEY section-8(f) rows (N1 sensor, N2 robust_control, N3 PPO update, N4 outer loops)
on one MI355X, each next to the fp64 C oracle of the same thing on the host cores -- one JSON line per row.
HIP-event timing on the launch stream; algorithmic bytes / flops per unit stated in each line."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import autonomous_quadrotor_environment_amd as aq
from autonomous_quadrotor_environment_amd.quadrotor_env import fast_random_init_states
from oracle import quad_oracle as O

HBM = 8000.0
O.build()
threads = max(1, min(len(os.sched_getaffinity(0)), 16))


def gpu_us(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def cpu_rate(work, units_per_call, seconds=3.0):
    """work(k) runs on thread k; returns units/s over all threads."""
    t0 = time.perf_counter(); work(0); t1 = time.perf_counter() - t0
    reps = max(1, int(seconds / max(t1, 1e-4)))
    def loop(k):
        for _ in range(reps):
            work(k)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(loop, range(threads)))
    return threads * reps * units_per_call / (time.perf_counter() - t0)


def out(**kw):
    print(json.dumps(kw), flush=True)


x0h = np.array([0, 0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0])

# ---- N1 sensor model: one sensor_sp per env and step --------------------------------------------------------------
N = 1 << 20
env = aq.BatchedQuad(N, 0.01, 10 ** 6, training=False, direct_control=1, T=1)
env.reset(torch.from_numpy(np.ascontiguousarray(fast_random_init_states(3, N).T)).float())
env.step(None)
sens = aq.BatchedSensor(env, seed=1)
sens.reset()
buf = torch.empty(24, N, device="cuda")
us = gpu_us(lambda: sens.sensor_sp(out=buf))
bytes_per = 4 * (13 + 3 + 4 + 2 * 25 + 24 + 4)  # state, accel, w, sensor rows r/w, outputs, quaternion rows written back
oe = O.OracleQuad(0.01, 10 ** 6, training=False, direct_control=1, T=1); oe.reset(x0h); oe.step(np.zeros(4))
osens = [O.OracleSensor(oe) for _ in range(threads)]
for s in osens:
    s.reset()
z = np.random.RandomState(0).normal(0, 1, 27)
def sens_work(k):
    for _ in range(2000):
        osens[k].sensor_sp(z)
cpu = cpu_rate(sens_work, 2000)
out(row="N1 sensor_sp (IMU+GPS+TRIAD+dead reckoning, Philox noise)", envs=N, us_per_launch=round(us, 1),
    reads_per_s=round(N / us * 1e6), algorithmic_bytes_per_read=bytes_per, hbm_GBps=round(N * bytes_per / us / 1e3, 1),
    hbm_frac=round(N * bytes_per / us / 1e3 / HBM, 3), cpu_oracle_reads_per_s=round(cpu), cpu_threads=threads)

# ---- N2 robust_control: fused constant-action rollout, interpolated-wind steps and a redraw step -----------------
N = 65536
x0 = torch.from_numpy(np.ascontiguousarray(fast_random_init_states(2, N).T)).float()
x0[0:5:2] = 0; x0[1:6:2] *= 0.05; x0[6] = 1; x0[7:10] = 0; x0[10:13] *= 0.05
for robust in (False, True):
    env = aq.BatchedQuad(N, 0.01, 10 ** 9, training=False, direct_control=1, T=1)
    env.robust_control = robust
    if robust:
        env.robust_gusts(None, seed=3); env.robust_params(None, seed=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.reset(x0)
    torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
    # full steps (robust refuses dynamics-only).  i = 1 (redraw) happens inside reset; launches cover i = 2..101 (warm-up),
    # 102..401 (timed: interpolated wind only), then 402..501 (timed separately: contains the redraw step i = 501)
    us = gpu_us(lambda: env.rollout("const", 100, flags=0), reps=3, warm=1)
    us_redraw = gpu_us(lambda: env.rollout("const", 100, flags=0), reps=1, warm=0)
    rec = dict(row="N2 robust_control rollout" if robust else "N2 reference point: nominal plant, same launch", envs=N,
               us_per_100_step_launch=round(us, 1), env_steps_per_s=round(N * 100 / us * 1e6),
               us_per_100_step_launch_containing_i501=round(us_redraw, 1), reset_ms=round(t_reset * 1e3, 2))
    if robust:
        rec["redraw_step_gust_draws_mean_max"] = [round(float(env.gust_evals.float().mean()), 1), int(env.gust_evals.max())]
        rng = np.random.RandomState(0)
        gd = rng.normal(0, 1, (4000, 3)) * [5, 5, 2]
        oes = []
        for k in range(threads):
            e = O.OracleQuad(0.01, 10 ** 6, training=False, direct_control=1, T=1)
            e.set_robust(True, params=np.r_[rng.random_sample(4) * .1, rng.normal(0, .3), rng.random_sample(4) * .1, rng.normal(0, .1, 3)],
                         gust_draws=gd)
            e.reset(x0h + np.r_[np.zeros(13)]); oes.append(e)
        def rob_work(k):
            for _ in range(400):
                oes[k].step(np.zeros(4))
        rec["cpu_oracle_env_steps_per_s"] = round(cpu_rate(rob_work, 400)); rec["cpu_threads"] = threads
    out(**rec)

# ---- N4 outer position loops: a policy step with the loop evaluated inside the step kernel -----------------------
N = 131072
w = aq.ActorCriticWeights.random(128, 75, 4, seed=1)
w = aq.ActorCriticWeights(w.actor, None)
env = aq.BatchedQuad(N, 0.01, 10 ** 6, training=False, direct_control=1, T=5)
env.reset(torch.from_numpy(np.ascontiguousarray(fast_random_init_states(4, N).T)).float(), return_history=False, push_window=True)
act = torch.zeros(4, N, device="cuda")
outs = (torch.empty(14, N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda", dtype=torch.uint8))
def pol_step():
    env.mlp_forward(w, "actor", out=act)
    env.step(act, out=outs, push_window=True)
us_plain = gpu_us(pol_step, reps=20)
tg = torch.zeros(3, N, device="cuda"); tg[2] = -5
env.outer_pospi(tg); env.outer_reset()
us_outer = gpu_us(pol_step, reps=20)
out(row="N4 policy step with pos_pi outer loop in the step kernel (k_mlp + k_step)", envs=N, us_per_step_plain=round(us_plain, 1),
    us_per_step_with_outer_loop=round(us_outer, 1), env_steps_per_s=round(N / us_outer * 1e6),
    added_algorithmic_bytes_per_step=4 * (3 + 2 * 3 + 3 + 3), note="the loop adds 3 target reads, 3+3 state rows r/w, 3 command rows")
