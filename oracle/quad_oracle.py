"""ctypes front-end of oracle/quad_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It mirrors the reference's duck-typed ``quad`` API
(environment/quadrotor_env.py:111-577) on top of the plain-C fp64 restatement so
parity tests can drive oracle and HIP path with the same calls.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libquad_oracle.so")

VARIANT_MAIN, VARIANT_OPT = 0, 1
INT_RK45, INT_DP5, INT_RK4 = 0, 1, 2
INTEGRATORS = {"rk45": INT_RK45, "dp5": INT_DP5, "rk4": INT_RK4}

_dp = C.POINTER(C.c_double)


_SO_F32 = os.path.join(_HERE, "_build", "libquad_oracle_f32.so")


def build(force=False):
    """Compile the oracle with gcc (seconds).  Safe to call repeatedly."""
    src = os.path.join(_HERE, "quad_oracle.c")
    srcs = [src, os.path.join(_HERE, "quad_oracle_f32.c"), os.path.join(_HERE, "Makefile")]
    if force or any(not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs)
                    for so in (_SO, _SO_F32)):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    return _SO


class _Env(C.Structure):
    _fields_ = [
        ("t_step", C.c_double), ("n", C.c_int), ("T", C.c_int), ("training", C.c_int),
        ("direct", C.c_int), ("clipped", C.c_int), ("variant", C.c_int), ("integrator", C.c_int),
        ("bb_cond", C.c_double * 9), ("zero_control", C.c_double * 4),
        ("previous_state", C.c_double * 13), ("state", C.c_double * 13),
        ("quat_state", C.c_double * 14),
        ("ang", C.c_double * 3), ("prev_ang", C.c_double * 3), ("ang_vel", C.c_double * 3),
        ("V_q", C.c_double * 4), ("w", C.c_double * 4), ("step_effort", C.c_double * 4),
        ("clipped_action", C.c_double * 4), ("action", C.c_double * 4), ("accel", C.c_double * 3),
        ("prev_shaping", C.c_double), ("has_prev_shaping", C.c_int),
        ("i", C.c_int), ("done", C.c_int), ("solved", C.c_int),
        ("reward", C.c_double), ("abs_sum", C.c_double),
        ("target_state", C.c_double), ("current_state", C.c_double),
        ("nfev", C.c_int), ("nsub", C.c_int),
        ("robust", C.c_int), ("ep_kf", C.c_double * 4), ("ep_m", C.c_double), ("ep_ir", C.c_double * 4),
        ("ep_J", C.c_double * 3), ("gust", C.c_double * 3), ("last_gust", C.c_double * 3),
        ("gust_draws", C.c_void_p), ("gust_cursor", C.c_int), ("gust_count", C.c_int),
    ]


class _Pid(C.Structure):
    _fields_ = [("p", C.c_double * 6), ("i", C.c_double * 6), ("d", C.c_double * 6),
                ("ix", C.c_double * 6), ("x_old", C.c_double * 6),
                ("ang_d_ant", C.c_double * 3), ("ts", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        assert _lib.qo_env_size() == C.sizeof(_Env), "oracle struct layout drifted"
        _lib.qo_batch_rollout.restype = C.c_long
        _lib.qo_batch_ppo.restype = C.c_long
    return _lib


def _a(x, n=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    if n is not None:
        assert x.size == n, (x.shape, n)
    return x


def _p(x):
    return x.ctypes.data_as(_dp)


# ---- functional wrappers ------------------------------------------------------
def euler_quat(ang):
    q = np.zeros(4)
    lib().qo_euler_quat(_p(_a(ang, 3)), _p(q))
    return q


def quat_euler(q):
    ang = np.zeros(3)
    lib().qo_quat_euler(_p(_a(q, 4)), _p(ang))
    return ang


def deriv_quat(w, q):
    dq = np.zeros(4)
    lib().qo_deriv_quat(_p(_a(w, 3)), _p(_a(q, 4)), _p(dq))
    return dq


def quat_rot_mat(q):
    R = np.zeros(9)
    lib().qo_quat_rot_mat(_p(_a(q, 4)), _p(R))
    return R.reshape(3, 3)


def f2F(a):
    w, F, M = np.zeros(4), C.c_double(), np.zeros(3)
    lib().qo_f2F(_p(_a(a, 4)), _p(w), C.byref(F), _p(M))
    return w, F.value, M


def f2w(f, m, clipped=True, variant=VARIANT_MAIN):
    se, w, F, M = np.zeros(4), np.zeros(4), C.c_double(), np.zeros(3)
    lib().qo_f2w(int(variant), int(bool(clipped)), C.c_double(float(f)), _p(_a(m, 3)), _p(se), _p(w),
                 C.byref(F), _p(M))
    return se, w, F.value, M


def lqr_gains(clipped=True):
    kt, katt = np.zeros(6), np.zeros(6)
    lib().qo_lqr_gains(int(bool(clipped)), _p(kt), _p(katt))
    return kt.reshape(3, 2), katt.reshape(3, 2)


def mlp_forward(x, W1, b1, W2, b2, W3, b3, final_tanh):
    x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    W1, b1, W2, b2, W3, b3 = (_a(v) for v in (W1, b1, W2, b2, W3, b3))
    h, in_dim = W1.shape
    out_dim = W3.shape[0]
    y = np.zeros((x.shape[0], out_dim))
    for r in range(x.shape[0]):
        xr = _a(x[r])
        lib().qo_mlp_forward(_p(xr), in_dim, h, out_dim, _p(W1), _p(b1), _p(W2), _p(b2), _p(W3), _p(b3),
                             int(bool(final_tanh)), _p(y[r]))
    return y


def gae(values, masks, rewards, ddof):
    values, masks, rewards = _a(values), _a(masks), _a(rewards)
    n = rewards.size
    assert values.size == n + 1 and masks.size == n
    ret, adv = np.zeros(n), np.zeros(n)
    lib().qo_gae(_p(values), _p(masks), _p(rewards), n, int(ddof), _p(ret), _p(adv))
    return ret, adv


def ppo_loss(mean, value, actions, old_logprob, adv, returns, std, eps=0.2):
    """[B,4] / [B] arrays -> (loss.mean(), mean actor, critic_loss, mean entropy term)."""
    mean, actions, old_logprob = _a(mean), _a(actions), _a(old_logprob)
    value, adv, returns = _a(value), _a(adv), _a(returns)
    out = np.zeros(4)
    lib().qo_ppo_loss(int(adv.size), _p(mean), _p(value), _p(actions), _p(old_logprob), _p(adv), _p(returns),
                      C.c_double(float(std)), C.c_double(float(eps)), _p(out))
    return out


def ppo_update(actor, critic, std, obs, actions, old_logprob_joint, adv, returns, batches, lr, betas=(0.9, 0.999), eps=1e-8,
               eps_clip=0.2, c_value=0.5, c_entropy=0.006, learn_std=False):
    """The optimisation loop both PPO.update variants share, in fp64 numpy (test infrastructure; small batches only):
    environment/controller/ppo.py:172-203 (full batch per epoch: `batches` = K_epochs x [all samples], c_entropy 0.006) and
    visual_landing/ppo_trainer.py:96-131,187-196 (`batches` = the minibatches of every epoch in order, c_entropy 0.01,
    learned std).  Per batch: evaluate (model.py:74-85 / rl_nn_model_v2.py:232-247: Normal(mean, std) per action dimension,
    joint log-probability = their sum, entropy = 4 (1/2 + log(2 pi)/2 + log std)), ratio = exp(logp - old logp), loss =
    mean(-min(ratio A, clip(ratio, 1 - eps_clip, 1 + eps_clip) A)) + c_value mean((V - R)^2) - c_entropy mean(entropy),
    backward, one torch.optim.Adam step over [actor, critic(, std)].
    actor / critic: [W1, b1, W2, b2, W3, b3] (torch layout), tanh hidden layers, tanh on the actor's output.
    Returns (actor, critic, std, [per-batch dict(actor=, critic=)])."""
    P = [np.array(v, dtype=np.float64) for v in list(actor) + list(critic)] + [np.array(float(std))]
    m = [np.zeros_like(v) for v in P]
    v2 = [np.zeros_like(v) for v in P]
    obs, actions, lpo, adv, returns = (np.asarray(a, dtype=np.float64) for a in (obs, actions, old_logprob_joint, adv, returns))
    log = []

    def mlp(x, W, final_tanh):
        h1 = np.tanh(x @ W[0].T + W[1]); h2 = np.tanh(h1 @ W[2].T + W[3]); o = h2 @ W[4].T + W[5]
        return h1, h2, (np.tanh(o) if final_tanh else o)

    def mlp_back(x, W, h1, h2, out, d_out, final_tanh):
        dz3 = d_out * (1 - out * out) if final_tanh else d_out
        dz2 = (dz3 @ W[4]) * (1 - h2 * h2)
        dz1 = (dz2 @ W[2]) * (1 - h1 * h1)
        return [dz1.T @ x, dz1.sum(0), dz2.T @ h1, dz2.sum(0), dz3.T @ h2, dz3.sum(0)]
    for t, idx in enumerate(batches, 1):
        x, a, A, R = obs[idx], actions[idx], adv[idx], returns[idx]
        B, sd = len(idx), float(P[12])
        h1, h2, mean = mlp(x, P[0:6], True)
        d = a - mean
        logp = (-0.5 * d * d / sd ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)).sum(1)
        ratio = np.exp(logp - lpo[idx])
        s1, s2 = ratio * A, np.clip(ratio, 1 - eps_clip, 1 + eps_clip) * A
        use1 = s1 <= s2                                                     # torch.min: surr1 on ties
        live = use1 | ((ratio > 1 - eps_clip) & (ratio < 1 + eps_clip))    # the clamp passes a gradient only inside the clip
        g_lp = np.where(live, -A * ratio, 0.0) / B                          # d mean(actor_loss) / d logp
        g = mlp_back(x, P[0:6], h1, h2, mean, g_lp[:, None] * d / sd ** 2, True)
        c1, c2, val = mlp(x, P[6:12], False)
        err = val[:, 0] - R
        g += mlp_back(x, P[6:12], c1, c2, val, (2 * c_value * err / B)[:, None], False)
        g_std = (g_lp * ((d * d / sd ** 2 - 1).sum(1) / sd)).sum() - c_entropy * 4.0 / sd
        g.append(np.array(g_std if learn_std else 0.0))
        log.append(dict(actor=float(np.mean(-np.minimum(s1, s2))), critic=float(c_value * np.mean(err * err))))
        for k in range(13):   # torch.optim.Adam (no amsgrad, no weight decay)
            if k == 12 and not learn_std:
                continue
            m[k] = betas[0] * m[k] + (1 - betas[0]) * g[k]
            v2[k] = betas[1] * v2[k] + (1 - betas[1]) * g[k] * g[k]
            P[k] = P[k] - lr / (1 - betas[0] ** t) * m[k] / (np.sqrt(v2[k]) / np.sqrt(1 - betas[1] ** t) + eps)
    return P[0:6], P[6:12], float(P[12]), log


def visual_reward(total_steps, marker, pos, vel, control, last_shaping, step, ang, v_ang):
    out = np.zeros(4)
    ls = 0.0 if last_shaping is None else float(last_shaping)
    lib().qo_visual_reward(int(total_steps), _p(_a(marker, 3)), _p(_a(pos, 3)), _p(_a(vel, 3)),
                           _p(_a(control, 3)), C.c_double(ls), int(step), _p(_a(ang, 3)), _p(_a(v_ang, 3)),
                           _p(out))
    return out[0], out[1], bool(out[2]), int(out[3])


def random_init_states(seed_or_rng, n_envs=1):
    """Replay of quad.reset()'s random branch draw order (quadrotor_env.py:426,440-445,
    robust_control.reset :97-101), one env after another on ONE legacy MT19937 stream
    (the reference uses the global np.random).  Returns float64 [n_envs, 13]."""
    rng = seed_or_rng if isinstance(seed_or_rng, np.random.RandomState) else np.random.RandomState(seed_or_rng)
    out = np.zeros((n_envs, 13))
    for k in range(n_envs):
        rng.random_sample(4); rng.normal(0, 0.3, 1); rng.random_sample(4)
        rng.normal(np.zeros(3), 0.1 * np.ones(3), [3])
        ang = rng.rand(3) - 0.5
        q = euler_quat(ang)
        out[k, 0:5:2] = np.clip(rng.normal([0, 0, 0], 2), -2.5, 2.5)
        out[k, 1:6:2] = np.clip(rng.normal([0, 0, 0], 2), -5.0, 5.0)
        out[k, 6:10] = q
        out[k, 10:13] = np.clip(rng.normal([0, 0, 0], 2), -15.0, 7.5)
    return out


_lib32 = None


def lib_f32():
    """The fp32 build (quad_oracle_f32.c): batch drivers only, float arrays."""
    global _lib32
    if _lib32 is None:
        build()
        _lib32 = C.CDLL(_SO_F32)
        _lib32.qo_batch_rollout.restype = C.c_long
        _lib32.qo_batch_ppo.restype = C.c_long
    return _lib32


def batch_rollout(states, steps, t_step=0.01, direct=1, variant=VARIANT_MAIN, integrator="rk45",
                  controller=0, act=None, fp32=False):
    """N independent envs advanced `steps` steps (after reset's single warm-up step).
    controller: 0 constant action, 1 LQR, 2 PID.  Returns (final states [N,13], reward sums [N]).
    fp32: the float build of the same source (CPU-baseline timing only)."""
    ft, ct, L = (np.float32, C.c_float, lib_f32()) if fp32 else (np.float64, C.c_double, lib())
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    s = np.array(states, dtype=ft, order="C").reshape(-1, 13)
    rs = np.zeros(s.shape[0], dtype=ft)
    act = np.ascontiguousarray(np.zeros(4) if act is None else act, dtype=ft)
    assert act.size == 4
    L.qo_batch_rollout(s.shape[0], int(steps), ct(t_step), int(direct), int(variant),
                       INTEGRATORS[integrator], int(controller), vp(act), vp(s), vp(rs))
    return s, rs


def batch_ppo(states, steps, actor, critic=None, T=5, t_step=0.01, integrator="rk45", fp32=False):
    """N envs with the actor (and optionally the critic) MLP in the loop; weights = lists
    [W1, b1, W2, b2, W3, b3] (torch layout).  Returns (final states [N,13], value sums [N])."""
    ft, ct, L = (np.float32, C.c_float, lib_f32()) if fp32 else (np.float64, C.c_double, lib())
    vp = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    s = np.array(states, dtype=ft, order="C").reshape(-1, 13)
    vs = np.zeros(s.shape[0], dtype=ft)
    a = [np.ascontiguousarray(w, dtype=ft) for w in actor]
    c = [np.ascontiguousarray(w, dtype=ft) for w in critic] if critic is not None else [None] * 6
    L.qo_batch_ppo(s.shape[0], int(steps), ct(t_step), INTEGRATORS[integrator], int(T), int(a[0].shape[0]),
                   *[vp(x) for x in a], *[vp(x) for x in c], vp(s), vp(vs))
    return s, vs


class OracleQuad:
    """Same call surface as reference ``quad`` (quadrotor_env.py:111-577).

    variant="opt" follows quadrotor_env_opt.py (``training`` then plays PPO_TRAINING).
    integrator: "rk45" = scipy-exact adaptive DP5(4) (reference behaviour),
    "dp5" = one fixed DP5 step, "rk4" = classic RK4.
    """

    def __init__(self, t_step, n, training=True, euler=0, direct_control=1, T=1, clipped=True,
                 variant="main", integrator="rk45"):
        self._e = _Env()
        self._variant = VARIANT_OPT if variant == "opt" else VARIANT_MAIN
        lib().qo_init(C.byref(self._e), C.c_double(t_step), int(n), int(bool(training)), int(bool(direct_control)),
                      int(T), int(bool(clipped)), self._variant, INTEGRATORS[integrator])
        self.T = T
        self.t_step = t_step
        self.mass, self.gravity = 1.03, 9.82
        self.J_mat = np.diag([16.83e-3, 16.83e-3, 28.34e-3])
        self.state_size, self.action_size = 13, 4
        self._rng = np.random.RandomState()

    def seed(self, seed):
        self._rng = np.random.RandomState(seed)

    # attribute views
    def _vec(self, name):
        return np.array(getattr(self._e, name)[:])

    state = property(lambda s: s._vec("state"))
    previous_state = property(lambda s: s._vec("previous_state"))
    quat_state = property(lambda s: s._vec("quat_state")[None, :])
    ang = property(lambda s: s._vec("ang"))
    prev_ang = property(lambda s: s._vec("prev_ang"))
    ang_vel = property(lambda s: s._vec("ang_vel"))
    w = property(lambda s: s._vec("w")[:, None])
    step_effort = property(lambda s: s._vec("step_effort"))
    clipped_action = property(lambda s: s._vec("clipped_action"))
    zero_control = property(lambda s: s._vec("zero_control"))
    bb_cond = property(lambda s: s._vec("bb_cond"))
    accel = property(lambda s: s._vec("accel")[:, None])
    i = property(lambda s: s._e.i)
    n = property(lambda s: s._e.n)
    done = property(lambda s: bool(s._e.done))
    solved = property(lambda s: s._e.solved)
    reward = property(lambda s: s._e.reward)
    abs_sum = property(lambda s: s._e.abs_sum)
    nfev = property(lambda s: s._e.nfev)
    nsub = property(lambda s: s._e.nsub)
    prev_shaping = property(lambda s: s._e.prev_shaping if s._e.has_prev_shaping else None)

    def drone_eq(self, t, x, action):
        out = np.zeros(13)
        lib().qo_drone_eq(C.byref(self._e), _p(_a(x, 13)), _p(_a(action, 4)), _p(out))
        return out

    def set_robust(self, on=True, params=None, gust_draws=None):
        """env.robust_control = on; params = the 12 episode draws (kf4, m, ir4, J3), gust_draws [n,3] = the
        np.random.normal(0,[5,5,2]) triples robust_control.wind consumes, in call order."""
        pp = _a(params, 12) if params is not None else None
        if gust_draws is not None:
            self._gd = _a(gust_draws).reshape(-1, 3).copy()
            lib().qo_set_robust(C.byref(self._e), int(bool(on)), _p(pp) if pp is not None else None,
                                _p(self._gd), int(self._gd.shape[0]))
        else:
            lib().qo_set_robust(C.byref(self._e), int(bool(on)), _p(pp) if pp is not None else None, None, -1)

    gust_cursor = property(lambda s: s._e.gust_cursor)
    gust = property(lambda s: s._vec("gust"))
    last_gust = property(lambda s: s._vec("last_gust"))

    def set_w(self, w):
        for k in range(4):
            self._e.w[k] = float(np.asarray(w).ravel()[k])

    def set_previous_state(self, x):
        x = _a(x, 13)
        for k in range(13):
            self._e.previous_state[k] = x[k]

    def reset(self, det_state=None):
        if det_state is None:
            if self._variant == VARIANT_OPT:  # no robust_parameters draws in the opt variant
                rng = self._rng
                ang = rng.rand(3) - 0.5
                s = np.zeros(13)
                s[0:5:2] = np.clip(rng.normal([0, 0, 0], 2), -2.5, 2.5)
                s[1:6:2] = np.clip(rng.normal([0, 0, 0], 2), -5.0, 5.0)
                s[6:10] = euler_quat(ang)
                s[10:13] = np.clip(rng.normal([0, 0, 0], 2), -15.0, 7.5)
            else:
                s = random_init_states(self._rng, 1)[0]
        else:
            s = _a(det_state, 13)
        states = np.zeros((self.T, 14))
        actions = np.zeros((self.T, 4))
        lib().qo_reset(C.byref(self._e), _p(s), _p(states), _p(actions))
        return states, actions

    def step(self, action):
        lib().qo_step(C.byref(self._e), _p(_a(action, 4)))
        return self.quat_state, self._e.reward, bool(self._e.done)


class OraclePid:
    """pid_control of environment/controller/pid_vel_control.py:29-110 (targets passed in)."""

    def __init__(self, env, clipped=True):
        self.env = env
        self._c = _Pid()
        lib().qo_pid_init(C.byref(self._c), int(bool(clipped)))

    def control(self, xd, dxd, psd, dpsd):
        a = np.zeros(4)
        lib().qo_pid_action(C.byref(self._c), C.byref(self.env._e), _p(_a(xd, 3)), C.c_double(float(psd)), _p(a))
        return a


class _Sensor(C.Structure):
    _fields_ = [("std", C.c_double * 5), ("a_b_d", C.c_double), ("g_b_d", C.c_double), ("m_b_d", C.c_double),
                ("a_b_accel", C.c_double), ("g_b", C.c_double), ("position_t0", C.c_double * 3),
                ("velocity_t0", C.c_double * 3), ("quaternion_t0", C.c_double * 4), ("R", C.c_double * 9),
                ("quat_alias", C.c_int)]


class OracleSensor:
    """sensor of environment/quadrotor_env.py:579-724 with injected noise (z27 standard normals per sensor_sp call,
    u3 uniform draws per reset, in the reference's draw order)."""

    def __init__(self, env):
        self.env = env
        self._s = _Sensor()
        assert lib().qo_sensor_size() == C.sizeof(_Sensor)
        lib().qo_sensor_init(C.byref(self._s))

    def reset(self, u3=(0.5, 0.5, 0.5), error=True):
        lib().qo_sensor_reset(C.byref(self._s), C.byref(self.env._e), _p(_a(u3, 3)), int(bool(error)))

    def sensor_sp(self, z27=None):
        """rl_worker.py:164-175.  Returns (states_sens[14], gps_pos[3], gps_vel[3], quaternion_triad[4])."""
        z = _a(np.zeros(27) if z27 is None else z27, 27)
        w = np.array(self.env._e.w[:])
        f_thrust = 1.435e-5 * float(np.sum(w * w))
        out = np.zeros(24)
        lib().qo_sensor_sp(C.byref(self._s), C.byref(self.env._e), C.c_double(f_thrust), _p(z), _p(out))
        return out[:14], out[14:17], out[17:20], out[20:24]


class OraclePidPos:
    """pid_control of environment/controller/PID_control.py:29-106 (position tracking)."""

    def __init__(self, env):
        self.env = env
        self._c = _Pid()
        lib().qo_pidpos_init(C.byref(self._c))

    def control(self, xd, dxd, psd, dpsd):
        a = np.zeros(4)
        lib().qo_pidpos_action(C.byref(self._c), C.byref(self.env._e), _p(_a(xd, 3)), _p(_a(dxd, 3)),
                               C.c_double(float(psd)), _p(a))
        return a


class OracleLqr:
    """Control law of environment/controller/lqr_quad.py:128-154."""

    def __init__(self, env, clipped=True):
        self.env = env
        self.kt, self.katt = lqr_gains(clipped)

    def control(self):
        a = np.zeros(4)
        lib().qo_lqr_action(C.byref(self.env._e), _p(_a(self.kt)), _p(_a(self.katt)), _p(a))
        return a


class OracleVelPid:
    """vel_pid of environment/controller/velocity_pid.py:16-35 with the caller-side state of position.py:209-210."""

    def __init__(self):
        self.cumm, self.prev_err = np.zeros(3), np.zeros(3)

    def __call__(self, state, target):
        out = np.zeros(14)
        lib().qo_vel_pid(_p(_a(state, 13)), _p(_a(target, 14)), _p(self.cumm), _p(self.prev_err), _p(out))
        return out


class OraclePosPi:
    """pos_pi of visual_landing/math_trajectory.py:46-62 plus the velocity clip of :123."""

    def __init__(self, P, D, I, dt):
        self.P, self.D, self.I, self.dt = float(P), float(D), float(I), float(dt)
        self.integrator = np.zeros(3)

    def vel_error(self, s, s_d, ds, ds_d):
        out = np.zeros(3)
        lib().qo_pos_pi(C.c_double(self.P), C.c_double(self.D), C.c_double(self.I), C.c_double(self.dt),
                        _p(self.integrator), _p(_a(s, 3)), _p(_a(s_d, 3)), _p(_a(ds, 3)), _p(_a(ds_d, 3)), _p(out))
        return out


class OracleWindow:
    """dl_in_gen of environment/controller/dl_auxiliary.py:15-32."""

    def __init__(self, T, state_size=13, action_size=4):
        self.T = T
        self.hist_size = (state_size - 3) + action_size + 1
        self.reset()

    def reset(self):
        self.deep_learning_input = np.zeros(self.hist_size * self.T, dtype=np.float32)

    def dl_input(self, states, actions):
        fp = C.POINTER(C.c_float)
        for s, a in zip(np.atleast_2d(states), np.atleast_2d(actions)):
            lib().qo_dl_input(self.deep_learning_input.ctypes.data_as(fp), self.T, _p(_a(s, 14)), _p(_a(a, 4)))
        return self.deep_learning_input


def reward_case(variant, training, state, ang, action, prev_shaping, i, done, n=100, T=1):
    """Drive done_condition()+reward_function() on a hand-set env (golden G7)."""
    env = OracleQuad(0.01, n, training=training, direct_control=1, T=T,
                     variant="opt" if variant else "main")
    e = env._e
    for k in range(13):
        e.state[k] = float(state[k])
    for k in range(3):
        e.ang[k] = float(ang[k])
    for k in range(4):
        e.action[k] = float(action[k])
    e.has_prev_shaping = 0 if prev_shaping is None or np.isnan(prev_shaping) else 1
    e.prev_shaping = 0.0 if not e.has_prev_shaping else float(prev_shaping)
    e.i = int(i)
    e.done = int(bool(done))
    e.solved = 0
    lib().qo_done_condition(C.byref(e))
    lib().qo_reward_function(C.byref(e))
    return e.reward, bool(e.done), e.solved, e.prev_shaping
