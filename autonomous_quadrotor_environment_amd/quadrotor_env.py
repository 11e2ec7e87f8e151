"""Host-side mirror of the reference environment API on top of libqenv.so.

* ``BatchedQuad`` -- N quadrotors, structure-of-arrays device tensors ([rows, N]), one HIP
  kernel launch per ``step()`` (or per fused ``rollout()``).
* ``quad``        -- the reference's duck-typed class for ONE env
  (/root/reference/environment/quadrotor_env.py:111-577): same constructor, ``seed``,
  ``reset``, ``step`` and the attributes controllers read after ``step``, so
  environment/controller/* scripts and visual_landing/ppo_trainer.py-style loops run unchanged.

Nothing here computes dynamics: every number comes from the HIP kernels.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

# constants callers read off the reference module (quadrotor_env.py:30-81)
M, G = 1.03, 9.82
T2WR = 2
J = np.array([[16.83e-3, 0, 0], [0, 16.83e-3, 0], [0, 0, 28.34e-3]])
BB_POS, BB_VEL, BB_ANG = 5, 10, np.pi / 2


_CTRL = {"const": L.CTRL_CONST, "lqr": L.CTRL_LQR, "pid": L.CTRL_PID, "pidpos": L.CTRL_PIDPOS}


def _raw_stream(device):
    """The caller's current HIP stream on `device` as an integer handle."""
    try:
        return torch._C._cuda_getCurrentRawStream(device.index)
    except AttributeError:  # (a torch without the raw accessor)
        return torch.cuda.current_stream(device).cuda_stream


def _current_device():
    try:
        return torch._C._cuda_getDevice()
    except AttributeError:
        return torch.cuda.current_device()


def euler_quat(ang):
    """quaternion_euler_utility.py:17-36 (host side, used only to build initial states)."""
    phi, theta, psi = (float(a) for a in ang)
    cp, sp = np.cos(phi / 2), np.sin(phi / 2)
    ct, st = np.cos(theta / 2), np.sin(theta / 2)
    cps, sps = np.cos(psi / 2), np.sin(psi / 2)
    q = np.array([cp * ct * cps + sp * st * sps, sp * ct * cps - cp * st * sps,
                  cp * st * cps + sp * ct * sps, cp * ct * sps - sp * st * cps])
    return q / np.linalg.norm(q)


def quat_euler(q):
    """quaternion_euler_utility.py:39-48 (host side convenience)."""
    q = np.asarray(q, dtype=np.float64).ravel()
    return np.array([np.arctan2(2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[1] ** 2 + q[2] ** 2)),
                     np.arcsin(2 * (q[0] * q[2] - q[3] * q[1])),
                     np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))])


def draw_robust_params(rng):
    """robust_control.reset (quadrotor_env.py:97-101) in the reference's draw order:
    episode_kf(4) ~ U(0,.1), episode_m ~ N(0,.3), episode_ir(4) ~ U(0,.1), diag episode_J(3) ~ N(0,.1)."""
    kf = rng.random_sample(4) * 0.1
    m = rng.normal(0, 0.3, 1)
    ir = rng.random_sample(4) * 0.1
    jj = rng.normal(np.zeros(3), 0.1 * np.ones(3), [3])
    return np.concatenate([kf, m, ir, jj])


def random_init_states(rng, n_envs, variant="main", return_params=False):
    """Initial 13-states in the reference's RNG draw order (reset(): quadrotor_env.py:426,
    440-445; robust_control.reset consumes 12 draws first, :97-101), env after env on one
    legacy MT19937 stream.  `rng`: np.random.RandomState, the np.random module, or a seed.
    Returns float64 [n_envs, 13] (and the [n_envs, 12] robust episode draws if asked)."""
    if isinstance(rng, (int, np.integer)):
        rng = np.random.RandomState(int(rng))
    out = np.zeros((n_envs, 13))
    params = np.zeros((n_envs, 12))
    for k in range(n_envs):
        if variant == "main":
            params[k] = draw_robust_params(rng)
        ang = rng.rand(3) - 0.5
        out[k, 0:5:2] = np.clip(rng.normal([0, 0, 0], 2), -BB_POS / 2, BB_POS / 2)
        out[k, 1:6:2] = np.clip(rng.normal([0, 0, 0], 2), -BB_VEL / 2, BB_VEL / 2)
        out[k, 6:10] = euler_quat(ang)
        out[k, 10:13] = np.clip(rng.normal([0, 0, 0], 2), -BB_VEL * 1.5, BB_POS * 1.5)
    return (out, params) if return_params else out


def _draw_block(g, n):
    """n initial states with the reference's distributions (quadrotor_env.py:440-445)."""
    ang = g.random((n, 3)) - 0.5
    out = np.zeros((n, 13))
    out[:, 0:5:2] = np.clip(g.normal(0, 2, (n, 3)), -BB_POS / 2, BB_POS / 2)
    out[:, 1:6:2] = np.clip(g.normal(0, 2, (n, 3)), -BB_VEL / 2, BB_VEL / 2)
    cp, sp = np.cos(ang[:, 0] / 2), np.sin(ang[:, 0] / 2)
    ct, st = np.cos(ang[:, 1] / 2), np.sin(ang[:, 1] / 2)
    cs, ss = np.cos(ang[:, 2] / 2), np.sin(ang[:, 2] / 2)
    q = np.stack([cp * ct * cs + sp * st * ss, sp * ct * cs - cp * st * ss, cp * st * cs + sp * ct * ss,
                  cp * ct * ss - sp * st * cs], 1)
    out[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
    out[:, 10:13] = np.clip(g.normal(0, 2, (n, 3)), -BB_VEL * 1.5, BB_POS * 1.5)
    return out


def fast_random_init_states(seed, n_envs, lo=0, hi=None, chunk=4096):
    """Vectorised initial states with the reference's distributions but NOT its draw order:
    env index -> (chunk, offset) with one Philox stream per chunk keyed by (seed, chunk), so
    any shard [lo, hi) of the global batch is bit-identical whatever the number of GPUs.
    Returns float64 [hi-lo, 13]."""
    hi = n_envs if hi is None else hi
    out = np.empty((hi - lo, 13))
    for c in range(lo // chunk, (hi - 1) // chunk + 1):
        g = np.random.Generator(np.random.Philox(key=np.array([int(seed), c], dtype=np.uint64)))
        block = _draw_block(g, chunk)
        a, b = max(lo, c * chunk), min(hi, (c + 1) * chunk)
        out[a - lo:b - lo] = block[a - c * chunk:b - c * chunk]
    return out


class BatchedQuad:
    """N independent quadrotors stepped by one HIP kernel launch.

    All tensors are structure-of-arrays on the device: state [13, N], actions [4, N],
    observations [14, N].  Constructor arguments follow the reference's
    ``quad(t_step, n, training, euler, direct_control, T, clipped)`` (quadrotor_env.py:112)
    after the leading ``n_envs``.
    """

    def __init__(self, n_envs, t_step, n, training=True, euler=0, direct_control=1, T=1, clipped=True,
                 variant="main", integrator="rk45", device="cuda", dtype=None, compute=None):
        """The handle.  Default (dtype=None, compute=None): the LEAN fp64 handle -- fp64 rows, fp64 multiply-adds,
        Newton-refined hardware reciprocals / square roots, guard-banded fast step-size control (include/qenv.h
        QENV_COMPUTE_LEAN) -- the parity-qualified one: it holds the 1e-5 tolerance against the reference's CPU step over
        every BASELINE config's horizon and is what bench.py measures.  Opt-in alternatives:
        dtype=torch.float64 (compute None / "native"): IEEE fp64 arithmetic; dtype=torch.float64, compute="f32": the
        mixed handle (fp64 rows, fp32 arithmetic); dtype=torch.float32: fp32 rows and arithmetic -- the fastest, but
        NOT parity-qualified (1e-4 off after 100 open-loop steps).  robust_control (domain randomisation) has kernels on
        the f32, mixed and IEEE fp64 handles only: `env.robust_control = True` on the lean default raises at once."""
        if dtype is None:
            dtype = torch.float64
            if compute is None:
                compute = "lean"
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be None (lean fp64 handle), torch.float32 or torch.float64")
        if compute not in (None, "native", "f32", "lean") or (compute in ("f32", "lean") and dtype != torch.float64):
            raise ValueError("compute must be None/'native', or 'f32' / 'lean' together with dtype=torch.float64")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedQuad runs on the GPU only (device='cuda[:i]'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_envs = int(n_envs)
        self.dtype = dtype
        self.compute = compute or "native"
        # 'lean' | 'f64' | 'mix' | 'f32' (bench.py's handle names)
        self.handle = {"lean": "lean", "f32": "mix"}.get(compute, "f64" if dtype == torch.float64 else "f32")
        self.t_step = t_step
        self.T = int(T)
        self.n = int(n) + self.T
        self.training = bool(training)
        self.direct_control_flag = direct_control
        self.clipped = clipped
        self.variant = variant
        self.integrator = integrator
        self.state_size, self.action_size = 13, 4
        self.mass, self.gravity = M, G
        self.J_mat = J
        self._cfg = L.QenvConfig(self.n_envs, float(t_step), int(n), self.T, int(bool(training)),
                                 int(bool(direct_control)), int(bool(clipped)), L.VARIANTS[variant],
                                 L.INTEGRATORS[integrator], L.F64 if dtype == torch.float64 else L.F32,
                                 {"f32": 1, "lean": 2}.get(compute, 0))
        nbytes = L.lib.qenv_workspace_bytes(C.byref(self._cfg))
        if nbytes == 0:
            raise L.QenvError("qenv_workspace_bytes rejected the configuration")
        with torch.cuda.device(self.device):
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._h = C.c_void_p()
            self._bound_stream = _raw_stream(self.device)   # the stream the handle launches on (see _bind)
            L.check(L.lib.qenv_create(C.byref(self._cfg), L.ptr(self._ws), nbytes, C.c_void_p(self._bound_stream), C.byref(self._h)),
                    "qenv_create")
        self._fields = {}
        self._order = None          # placement (regroup): logical env index of every SLOT of the batch, None = natural order
        self._inv = None            # ... and the slot of every logical env
        if direct_control:
            self.zero_control = np.ones(4) * (2 / T2WR - 1)
        else:
            self.zero_control = np.array([M * G, 0, 0, 0])
        if variant == "opt":
            bb = np.array([10, 10, 10, np.pi / 2, np.pi / 2, 4, 30, 30, 30], dtype=float)
            self.bb_cond = bb if training else bb * 1000
        else:
            self.bb_cond = np.array([10, 10, 10, np.pi / 2, np.pi / 2, 3 / 4 * np.pi, 20, 20, 20])

    # ---- plumbing -------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(_raw_stream(self.device))

    def _bind(self):
        # the library launches on the caller's current stream of ITS device: make that device current for this thread
        # (kernels launched into a stream of another device than the current one fail).  This runs in front of every library
        # call: the raw-handle lookups (0.3 us instead of 1.8 for torch.cuda.current_stream(...).cuda_stream) and skipping the
        # qenv_set_stream call while the stream has not changed take the wrapper from 3.4 to ~1 us per call -- 2-3 % of a
        # 20-step launch as the driver times it
        idx = self.device.index
        if _current_device() != idx:
            torch.cuda.set_device(self.device)
        st = _raw_stream(self.device)
        if st != self._bound_stream:
            L.check(L.lib.qenv_set_stream(self._h, C.c_void_p(st)), "qenv_set_stream")
            self._bound_stream = st

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and L is not None and getattr(L, "lib", None) is not None:  # module globals may be gone at exit
            L.lib.qenv_destroy(h)
            self._h = None

    def _field(self, fid, dtype=None):
        if fid not in self._fields:
            p, rows = C.c_void_p(), C.c_int32()
            L.check(L.lib.qenv_field(self._h, fid, C.byref(p), C.byref(rows)), "qenv_field")
            dt = dtype or self.dtype
            esz = torch.empty(0, dtype=dt).element_size()
            off = p.value - self._ws.data_ptr()
            nb = rows.value * self.n_envs * esz
            self._fields[fid] = self._ws[off:off + nb].view(dt).view(rows.value, self.n_envs)
        return self._fields[fid]

    def _as_rows(self, x, rows, name):
        """Accept [rows, N] (native) or [N, rows]; return a contiguous device tensor [rows, N]."""
        if x is None:
            return None
        if not isinstance(x, torch.Tensor):  # numpy arrays, lists, tuples
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
        x = x.to(device=self.device, dtype=self.dtype)
        if x.dim() == 1 and x.numel() == rows:
            x = x.view(rows, 1).expand(rows, self.n_envs)
        if x.shape == (rows, self.n_envs):
            return x.contiguous()
        if x.shape == (self.n_envs, rows):
            return x.t().contiguous()
        raise ValueError(f"{name}: expected shape [{rows}, {self.n_envs}] (or its transpose), got {tuple(x.shape)}")

    def _mask(self, mask):
        """uint8 [N] device copy of a reset / step mask (None passes through).  The kernels index it with the env number."""
        if mask is None:
            return None
        if not isinstance(mask, torch.Tensor):
            mask = torch.as_tensor(np.asarray(mask))
        if mask.numel() != self.n_envs:
            raise ValueError(f"mask: expected {self.n_envs} elements, got {tuple(mask.shape)}")
        m = mask.reshape(-1).to(device=self.device).to(torch.uint8)
        if self._order is not None:   # the caller's masks are in logical order
            m = m.index_select(0, self._order)
        return m.contiguous()

    # ---- placement: which lane steps which env ---------------------------------------------------
    # A wavefront runs as many RK45 attempts per env step as its slowest lane needs.  With the reference's random initial
    # states ~5 % of the envs are never stabilised by the LQR: they tumble at tens of rad/s and need two or three sub-steps per
    # env step for the rest of the episode -- and in the batch's natural order nearly every wave of 64 holds one (config 3: 1.73
    # attempts per wave-step where the lanes need 1.02: the census of tests/tools/regroup_experiment.py).  regroup() sorts the
    # envs by a key (default: the body-rate norm of their current states) so that the hard ones share waves: 1.25 attempts per
    # wave-step, +14 % env-steps/s on config 3, every env's trajectory bit for bit what it was (an env never sees its
    # neighbours).  The library moves every per-env row of the handle (qenv_permute_envs: also what a reset leaves alone --
    # prev_ang, ang_vel, PID state, banked returns); from then on every per-env view of this object (state, reward, flags,
    # episode_return ...) is in SLOT order -- logical(t) returns it in the caller's order, and reset() / masks are taken in
    # the caller's order.  step() with per-env actions refuses a regrouped batch (the rollouts with an on-device controller are
    # what this is for); clear_placement() moves everything back.
    def regroup(self, key=None):
        """Sort the envs by `key`, largest first ([N], one value per slot of the current placement; default / "rate": |omega| of
        the current states; "spin": |wz| + 0.3 |wxy|, the better predictor before an episode), now.  Returns the new order
        (logical env index per slot)."""
        if key is None or (isinstance(key, str) and key == "rate"):
            key = torch.linalg.vector_norm(self.state[10:13], dim=0)
        elif isinstance(key, str) and key == "spin":
            # BEFORE an episode (right after reset) the yaw rate predicts best which envs the LQR will not catch (its yaw
            # authority saturates): |wz| + 0.3 |wxy| of the initial states gives 1.40 attempts per wave-step on config 3
            # (|omega|: 1.49; the order drawn: 1.73; |omega| after an episode: 1.25)
            key = self.state[12].abs() + 0.3 * torch.linalg.vector_norm(self.state[10:12], dim=0)
        key = torch.as_tensor(key, device=self.device).reshape(-1)
        if key.numel() != self.n_envs:
            raise ValueError(f"regroup: key needs {self.n_envs} elements, got {key.numel()}")
        # the hard envs FIRST: a launch dispatches its workgroups in slot order, so the slow waves start first and the easy
        # ones fill the tail of the launch (ascending: the slowest waves are the last to start -- 2.4 % less on config 3)
        return self._permute(torch.argsort(key, descending=True, stable=True))

    def clear_placement(self):
        """Back to the natural order (every per-env row moves back)."""
        if self._inv is not None:
            self._permute(self._inv)

    def _permute(self, perm):
        """Slot p receives what slot perm[p] holds."""
        if getattr(self, "_sensor", None) is not None:
            raise L.QenvError("regroup(): not with an attached sensor (its per-env state lives outside the handle)")
        scratch = torch.empty(self._ws.numel(), dtype=torch.uint8, device=self.device)
        p32 = perm.to(torch.int32).contiguous()
        self._bind()
        L.check(L.lib.qenv_permute_envs(self._h, L.ptr(p32), L.ptr(scratch), scratch.numel()), "qenv_permute_envs")
        order = perm if self._order is None else self._order.index_select(0, perm)
        if torch.equal(order, torch.arange(self.n_envs, device=self.device)):
            self._order = self._inv = None
        else:
            self._order = order.contiguous()
            self._inv = torch.empty_like(order)
            self._inv[order] = torch.arange(self.n_envs, device=self.device)
        return order

    @property
    def slot_order(self):
        """[N] logical env index of every slot (None: natural order)."""
        return self._order

    def logical(self, t):
        """A per-env tensor of this object ([..., N], slot order) in the caller's env order (a copy when a placement is on)."""
        return t if self._inv is None else t.index_select(-1, self._inv)

    # ---- attributes (device views into the workspace, [rows, N]) -----------------------------
    state = property(lambda s: s._field(L.FIELD_STATE))
    previous_state = state
    ang = property(lambda s: s._field(L.FIELD_ANG))
    prev_ang = property(lambda s: s._field(L.FIELD_PREV_ANG))
    ang_vel = property(lambda s: s._field(L.FIELD_ANG_VEL))
    V_q = property(lambda s: s._field(L.FIELD_VQ))
    w = property(lambda s: s._field(L.FIELD_W))
    step_effort = property(lambda s: s._field(L.FIELD_STEP_EFFORT))
    clipped_action = property(lambda s: s._field(L.FIELD_CLIPPED_ACTION))
    accel = property(lambda s: s._field(L.FIELD_ACCEL))
    prev_shaping = property(lambda s: s._field(L.FIELD_PREV_SHAPING)[0])
    reward = property(lambda s: s._field(L.FIELD_REWARD)[0])
    abs_sum = property(lambda s: s._field(L.FIELD_ABS_SUM)[0])
    episode_return = property(lambda s: s._field(L.FIELD_EP_RETURN)[0])
    pid_state = property(lambda s: s._field(L.FIELD_PID))
    i = property(lambda s: s._field(L.FIELD_I, torch.int32)[0])
    flags = property(lambda s: s._field(L.FIELD_FLAGS, torch.int32)[0])
    window = property(lambda s: s._field(L.FIELD_WINDOW, torch.float32))
    last_return = property(lambda s: s._field(L.FIELD_LAST_RETURN)[0])
    episodes = property(lambda s: s._field(L.FIELD_EPISODES, torch.int32)[0])
    robust_parameters = property(lambda s: s._field(L.FIELD_ROBUST_PARAMS))  # [12, N] kf4, m, ir4, J3
    gust = property(lambda s: s._field(L.FIELD_GUST))                        # [6, N] gust(3), last_gust(3)
    gust_evals = property(lambda s: s._field(L.FIELD_GUST_EVALS, torch.int32)[0])
    done = property(lambda s: (s.flags & L.FLAG_DONE) != 0)
    solved = property(lambda s: ((s.flags & L.FLAG_SOLVED) != 0).to(torch.int32))

    @property
    def quat_state(self):
        return torch.cat([self.state[0:10], self.V_q], 0)

    # ---- reference API ----------------------------------------------------------------------
    def seed(self, seed):
        self._rng = np.random.RandomState(seed)

    def reset(self, det_state=None, mask=None, return_history=True, push_window=False):
        """quad.reset (quadrotor_env.py:408-454).  det_state: [13, N] / [N, 13] / [13] tensor
        or array; None draws every env's state in the reference's RNG order on the host.
        mask: uint8/bool [N], only those envs are reset (their episode return is banked in
        `last_return`).  push_window: the T warm-up slices refill the observation ring.
        Returns (states [T, 14, N], actions [T, 4, N])."""
        if det_state is None:
            rng = getattr(self, "_rng", None) or np.random.RandomState()
            self._rng = rng
            det_state = random_init_states(rng, self.n_envs, self.variant).T.copy()
        init = self._as_rows(det_state, 13, "det_state")
        if self._order is not None:   # det_state is in the caller's order
            init = init.index_select(1, self._order)
        m = self._mask(mask)
        states = torch.empty(self.T, 14, self.n_envs, device=self.device, dtype=self.dtype) if return_history else None
        actions = torch.empty(self.T, 4, self.n_envs, device=self.device, dtype=self.dtype) if return_history else None
        if return_history and mask is not None:
            states.zero_(); actions.zero_()
        self._bind()
        L.check(L.lib.qenv_reset(self._h, L.ptr(init), L.ptr(m), L.ptr(states), L.ptr(actions),
                                 int(bool(push_window))), "qenv_reset")
        return states, actions

    def reset_sampled(self, seed, episode=0, env_offset=0, mask=None, push_window=False):
        """Masked auto-reset with device-drawn initial states (sample_init_states + reset(mask) in one launch)."""
        if self._order is not None:
            raise L.QenvError("reset_sampled(): device-drawn initial states are keyed by the slot index -- not with regroup()")
        m = self._mask(mask)
        self._bind()
        L.check(L.lib.qenv_reset_sampled(self._h, int(seed), int(episode), int(env_offset), L.ptr(m),
                                         int(bool(push_window))), "qenv_reset_sampled")

    def step(self, action=None, flags=L.STEP_FULL, out=None, out_mask=None, push_window=False):
        """quad.step for all envs.  action [4, N] (or [N, 4]); None = zero_control.
        out: optional preallocated (obs [14, N] | None, reward [N] | None, done uint8 [N] | None);
        out_mask: optional [N] tensor receiving 1 - done; push_window: also append this step's
        [action, vel, q, q_dot] slice to the observation ring (dl_in_gen.dl_input).
        Returns (quat_state [14, N], reward [N], done [N] bool)."""
        if self._order is not None and action is not None and np.ndim(action) > 1:
            raise L.QenvError("step(): this batch is regrouped (regroup()): per-env actions would be taken in slot order -- "
                              "index them with env.slot_order yourself, or clear_placement() and reset")
        act = self._as_rows(action, 4, "action")
        self._bind()
        if push_window:
            flags |= L.STEP_PUSH_WINDOW
            if getattr(self, "_sensor", None) is not None and getattr(self, "_outer_kind", None) != "pospi":
                self._sensor.calls += 1  # mirrors the handle's count of fused sensor evaluations (BatchedSensor.attach)
        if (flags & 3) == L.STEP_DYNAMICS_ONLY:
            L.check(L.lib.qenv_step(self._h, L.ptr(act), None, None, None, None, flags), "qenv_step")
            return None, None, None
        if out is None:
            obs = torch.empty(14, self.n_envs, device=self.device, dtype=self.dtype)
            rew = torch.empty(self.n_envs, device=self.device, dtype=self.dtype)
            done = torch.empty(self.n_envs, device=self.device, dtype=torch.uint8)
        else:
            obs, rew, done = out
            self._check_buf(obs, 14, "out[0] (obs)")
            self._check_buf(rew, 0, "out[1] (reward)")
            self._check_buf(done, 0, "out[2] (done)", torch.uint8)
        self._check_buf(out_mask, 0, "out_mask")
        L.check(L.lib.qenv_step(self._h, L.ptr(act), L.ptr(obs), L.ptr(rew), L.ptr(done), L.ptr(out_mask), flags),
                "qenv_step")
        return obs, rew, (done.bool() if out is None else done)

    def rollout(self, controller="const", n_steps=1, action=None, pid_first_action=False, flags=L.STEP_FULL):
        """n_steps fused env steps in one launch with an on-device controller
        ('const' | 'lqr' | 'pid'); state stays in registers between steps."""
        ctrl = _CTRL[controller]
        ca = None
        if action is not None:  # CONST: the 4-vector action; PIDPOS: the 7-vector target
            vals = [float(a) for a in np.asarray(action).ravel()]
            ca = (C.c_double * len(vals))(*vals)
        self._bind()
        rc = L.lib.qenv_rollout(self._h, ctrl, n_steps, ca, 1 if pid_first_action else 0, flags)
        if rc:
            L.check(rc, "qenv_rollout")

    def track(self, targets, flags=L.STEP_FULL):
        """Position-PD trajectory tracking in one launch (PID_control.py:245-263): targets [n_steps, 7] =
        {xd, yd, zd, dxd, dyd, dzd, psi_d} per step, e.g. mission.table(n)."""
        if isinstance(targets, np.ndarray):
            targets = torch.from_numpy(np.ascontiguousarray(targets))
        tg = targets.to(device=self.device, dtype=self.dtype).contiguous()
        assert tg.dim() == 2 and tg.shape[1] == 7
        self._bind()
        L.check(L.lib.qenv_rollout_track(self._h, tg.shape[0], L.ptr(tg), flags), "qenv_rollout_track")

    def _rows_n(self, x, rows, name):
        """[rows, n] (or [n, rows]) array / tensor -> contiguous device tensor [rows, n] in the handle's dtype."""
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
        x = x.to(device=self.device, dtype=self.dtype)
        if x.dim() == 1:
            x = x.view(rows, -1) if rows > 1 else x.view(1, -1)
        if x.shape[0] != rows and x.shape[1] == rows:
            x = x.t()
        if x.dim() != 2 or x.shape[0] != rows:
            raise ValueError(f"{name}: expected [{rows}, n], got {tuple(x.shape)}")
        return x.contiguous()

    def reward_function(self, state, ang, action, prev_shaping, i, done):
        """quad.done_condition() + quad.reward_function() (quadrotor_env.py:500-573 / quadrotor_env_opt.py:410-480) on
        hand-set attributes for n samples: state [13, n], ang [3, n], action [4, n], prev_shaping [n] (NaN = None),
        i [n], done [n]; the handle supplies variant / training / n / the done box.  The device function of the step
        kernels.  Returns (reward [n], done bool [n], solved int32 [n], shaping [n])."""
        st, an, ac = self._rows_n(state, 13, "state"), self._rows_n(ang, 3, "ang"), self._rows_n(action, 4, "action")
        n = st.shape[1]
        ps = self._rows_n(prev_shaping, 1, "prev_shaping")
        ii = torch.as_tensor(np.asarray(i) if not isinstance(i, torch.Tensor) else i).to(device=self.device, dtype=torch.int32).contiguous()
        d0 = torch.as_tensor(np.asarray(done) if not isinstance(done, torch.Tensor) else done).to(device=self.device).to(torch.uint8).contiguous()
        if not (an.shape[1] == ac.shape[1] == ps.shape[1] == ii.numel() == d0.numel() == n):
            raise ValueError("reward_function: the inputs disagree on n")
        rew = torch.empty(n, device=self.device, dtype=self.dtype)
        sh = torch.empty_like(rew)
        dn = torch.empty(n, device=self.device, dtype=torch.uint8)
        so = torch.empty(n, device=self.device, dtype=torch.int32)
        self._bind()
        L.check(L.lib.qenv_reward_function(self._h, n, L.ptr(st), L.ptr(an), L.ptr(ac), L.ptr(ps), L.ptr(ii), L.ptr(d0),
                                           L.ptr(rew), L.ptr(dn), L.ptr(so), L.ptr(sh)), "qenv_reward_function")
        return rew, dn.bool(), so, sh

    def mix(self, action):
        """quad.f2F (direct handle, quadrotor_env.py:247-272) / quad.f2w (indirect, :197-245) for n actions [4, n]
        through the step kernels' action pre-processing.  Returns (w [4, n], [F, Mx, My, Mz] [4, n], step_effort [4, n])."""
        ac = self._rows_n(action, 4, "action")
        n = ac.shape[1]
        w, fm, eff = (torch.empty(4, n, device=self.device, dtype=self.dtype) for _ in range(3))
        self._bind()
        L.check(L.lib.qenv_mix(self._h, n, L.ptr(ac), L.ptr(w), L.ptr(fm), L.ptr(eff)), "qenv_mix")
        return w, fm, eff

    def lqr_action(self, out=None):
        out = torch.empty(4, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        self._check_buf(out, 4, "out")
        self._bind()
        L.check(L.lib.qenv_lqr_action(self._h, L.ptr(out)), "qenv_lqr_action")
        return out

    def pid_action(self, targets=None, out=None):
        out = torch.empty(4, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        tg = None if targets is None else (C.c_double * 4)(*[float(a) for a in targets])
        self._check_buf(out, 4, "out")
        self._bind()
        L.check(L.lib.qenv_pid_action(self._h, tg, L.ptr(out)), "qenv_pid_action")
        return out

    def pidpos_action(self, targets=None, out=None):
        """PID_control.py position PD: targets = [xd, yd, zd, dxd, dyd, dzd, psi_d]."""
        out = torch.empty(4, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        tg = None if targets is None else (C.c_double * 7)(*[float(a) for a in targets])
        self._check_buf(out, 4, "out")
        self._bind()
        L.check(L.lib.qenv_pidpos_action(self._h, tg, L.ptr(out)), "qenv_pidpos_action")
        return out

    def pid_reset(self, mask=None):
        m = self._mask(mask)
        self._bind()
        L.check(L.lib.qenv_pid_reset(self._h, L.ptr(m)), "qenv_pid_reset")

    # ---- PPO inner loop -----------------------------------------------------------------------
    def window_push(self, obs, actions, reset_first=False):
        o = self._as_rows(obs, 14, "obs")
        a = self._as_rows(actions, 4, "actions")
        self._bind()
        L.check(L.lib.qenv_window_push(self._h, L.ptr(o), L.ptr(a), int(bool(reset_first))), "qenv_window_push")

    def _check_head(self, W, name, n_out=None):
        """The kernels take raw pointers: a weight set of another window length, dtype or device would be read out of
        bounds (or mis-strided), so refuse it here.  W = [W1, b1, W2, b2, W3, b3] in torch layout."""
        W1, b1, W2, b2, W3, b3 = W
        H, K = int(W1.shape[0]), 15 * self.T
        want = [(H, K), (H,), (H, H), (H,), (int(W3.shape[0]), H), (int(W3.shape[0]),)]
        for t, shp, nm in zip(W, want, ("W1", "b1", "W2", "b2", "W3", "b3")):
            if tuple(t.shape) != shp:
                raise ValueError(f"{name}.{nm}: expected shape {shp} for a T={self.T} window (state_dim {K}), got {tuple(t.shape)}")
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{name}.{nm}: must be a contiguous float32 tensor on {self.device}")
        if n_out is not None and int(W3.shape[0]) != n_out:
            raise ValueError(f"{name}: expected {n_out} outputs, got {int(W3.shape[0])}")
        return H

    def _check_buf(self, t, rows, name, dtype=None):
        if t is None:
            return
        dtype = dtype or self.dtype
        shp = (rows, self.n_envs) if rows else (self.n_envs,)
        ok_shape = tuple(t.shape) == shp or (rows == 1 and tuple(t.shape) in ((self.n_envs,), (1, self.n_envs)))
        if not ok_shape or t.dtype != dtype or not t.is_contiguous() or t.device != self.device:
            raise ValueError(f"{name}: expected a contiguous {dtype} tensor of shape {shp} on {self.device}, got "
                             f"{tuple(t.shape)} {t.dtype} on {t.device}")

    @staticmethod
    def _prep_key(weights, head, W):
        """What a prepared image was built from: where the first layer lives and the version of THAT STORAGE
        (ActorCriticWeights.version_of: shared by every wrapper of the same tensors, so an alias held by this env sees the
        learner's touch()); a weights object without versions is keyed by its identity."""
        ver = weights.version_of(head) if hasattr(weights, "version_of") else (id(weights), getattr(weights, "version", 0))
        return (W[0].data_ptr(), ver)

    def _prepared_is_current(self, weights, head):
        """True if nothing would be launched by _ensure_prepared (no prepared form for this shape, or the image is current)."""
        W = weights.head(head)
        if self.T != 5 or L.lib.qenv_mlp_image_bytes(int(W[0].shape[1]), int(W[0].shape[0])) == 0:
            return True
        cur = self.__dict__.get("_prep", {}).get(0 if head == "actor" else 1)
        return cur is not None and cur[0] == self._prep_key(weights, head, W)

    def _ensure_prepared(self, weights, head):
        """Build / refresh this handle's prepared image of `weights.head(head)` (qenv_mlp_prepare: the weights in the
        forward kernel's LDS layout) when the weight object or its version changed.  Only the 75-128-128-n policy on a
        T = 5 handle has one; everything else stages the weights inside every launch."""
        W = weights.head(head)
        nbytes = L.lib.qenv_mlp_image_bytes(int(W[0].shape[1]), int(W[0].shape[0]))
        if nbytes == 0 or self.T != 5:
            return
        slot = 0 if head == "actor" else 1
        prep = self.__dict__.setdefault("_prep", {})
        key = self._prep_key(weights, head, W)
        cur = prep.get(slot)
        if cur is not None and cur[0] == key:
            return
        self._check_head(W, head)
        img = cur[1] if cur is not None else torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._bind()
        L.check(L.lib.qenv_mlp_prepare(self._h, slot, int(W[0].shape[1]), int(W[0].shape[0]), int(W[4].shape[0]),
                                       *[L.ptr(t) for t in W], L.ptr(img)), "qenv_mlp_prepare")
        prep[slot] = (key, img, weights)   # the image (and the weights it was built from) stay alive with the handle

    def mlp_forward(self, weights, head="actor", out=None, obs_store=None):
        """ActorCritic.actor / .critic forward on the observation ring (model.py:27-42).
        obs_store: optional float32 [15*T, N] receiving the window in time order."""
        W1, b1, W2, b2, W3, b3 = weights.head(head)
        self._check_head((W1, b1, W2, b2, W3, b3), head)
        n_out = W3.shape[0]
        out = torch.empty(n_out, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        self._check_buf(out, n_out, "out")
        self._check_buf(obs_store, 15 * self.T, "obs_store", torch.float32)
        self._ensure_prepared(weights, head)
        self._bind()
        L.check(L.lib.qenv_mlp_forward(self._h, W1.shape[1], W1.shape[0], n_out, int(head == "actor"), L.ptr(W1), L.ptr(b1),
                                       L.ptr(W2), L.ptr(b2), L.ptr(W3), L.ptr(b3), L.ptr(out), L.ptr(obs_store)),
                "qenv_mlp_forward")
        return out

    def policy_act(self, weights, noise=None, seed=0, step=0, env_offset=0, out=None, logprob=None, mean=None,
                   obs_store=None):
        """ActorCritic.act (model.py:52-72) in one launch: actor forward on the observation ring + Normal(mean, std)
        sample + per-dimension log-prob (mlp_forward followed by sample_action, bit for bit)."""
        W1, b1, W2, b2, W3, b3 = weights.head("actor")
        self._check_head((W1, b1, W2, b2, W3, b3), "actor", n_out=4)
        out = torch.empty(4, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        nz = None if noise is None else self._as_rows(noise, 4, "noise")
        for t, nm in ((out, "out"), (logprob, "logprob"), (mean, "mean")):
            self._check_buf(t, 4, nm)
        self._check_buf(obs_store, 15 * self.T, "obs_store", torch.float32)
        self._ensure_prepared(weights, "actor")
        self._bind()
        L.check(L.lib.qenv_policy_act(self._h, W1.shape[1], W1.shape[0], L.ptr(W1), L.ptr(b1), L.ptr(W2), L.ptr(b2), L.ptr(W3), L.ptr(b3),
                                      float(weights.std), L.ptr(nz), int(seed), int(step), int(env_offset), L.ptr(mean),
                                      L.ptr(out), L.ptr(logprob), L.ptr(obs_store)), "qenv_policy_act")
        return out

    def ppo_step(self, weights, seed, step, env_offset, actions, logprob, mean, obs_store, reward, done, mask, noise=None,
                 valid=None, inband=False, key_by_episode=False, obs_rows_from=0):
        """One step of the rollout worker loop with auto-reset (qenv_ppo_step): policy_act + step(NO_AUX, push_window) +
        reset_sampled(mask=done), bit-identical buffers, with the reset envs' warm-up overlapped with the next step (it
        stays PENDING after the call: ppo_flush(), or any other call that touches the env, completes it).
        obs_rows_from (in-band schedule, hidden 64 / 128, T = 5): only rows >= it of the window are written to obs_store --
        15 (T - 1) stores the newest slice alone (the slice layout of PolicyRollout, include/qenv.h)."""
        key = (id(weights), weights.actor[0].data_ptr())
        if getattr(self, "_ppo_checked", None) != key:  # validate a weight set once, not on every step
            self._check_head(weights.head("actor"), "actor", n_out=4)
            self._ppo_checked = key
        W = weights.head("actor")
        for t, nm in ((actions, "actions"), (logprob, "logprob"), (mean, "mean")):
            self._check_buf(t, 4, nm)
        self._check_buf(obs_store, 15 * self.T, "obs_store", torch.float32)
        for t, nm in ((reward, "reward"), (mask, "mask"), (valid, "valid")):
            self._check_buf(t, 0, nm)
        self._check_buf(done, 0, "done", torch.uint8)
        nz = None if noise is None else self._as_rows(noise, 4, "noise")
        self._ensure_prepared(weights, "actor")
        a = L.PpoStepArgs(int(W[0].shape[1]), int(W[0].shape[0]), *[t.data_ptr() for t in W], float(weights.std),
                          None if nz is None else nz.data_ptr(), int(seed), int(step), int(env_offset),
                          None if mean is None else mean.data_ptr(), actions.data_ptr(),
                          None if logprob is None else logprob.data_ptr(), None if obs_store is None else obs_store.data_ptr(),
                          reward.data_ptr(), done.data_ptr(), None if mask is None else mask.data_ptr(),
                          None if valid is None else valid.data_ptr(), int(bool(inband)), int(bool(key_by_episode)),
                          int(obs_rows_from), 0)
        self._bind()
        L.check(L.lib.qenv_ppo_step(self._h, C.byref(a)), "qenv_ppo_step")

    def policy_rollout(self, weights, n_steps, actions, reward=None, done=None, mask=None):
        """The evaluation loop of ppo_quad_eval.py:44-62 for n_steps steps in ONE library call (qenv_policy_rollout): per step
        the deterministic policy (mlp_forward of the actor) and step(NO_AUX, push_window) on its action; the same buffers as
        those 2 n_steps calls, bit for bit.  actions [n_steps, 4, N]; reward / mask [n_steps, N] optional; done [N] uint8
        optional (the last step's).  For the reference's controller (hidden 128, T = 5, RK45) the whole loop is one launch."""
        n_steps = int(n_steps)
        W = weights.head("actor")
        self._check_head(W, "actor", n_out=4)
        want = {"actions": (actions, (n_steps, 4, self.n_envs)), "reward": (reward, (n_steps, self.n_envs)),
                "mask": (mask, (n_steps, self.n_envs))}
        for nm, (t, shape) in want.items():
            if t is None and nm != "actions":
                continue
            if t is None or tuple(t.shape) != shape or t.dtype != self.dtype or not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"{nm}: expected a contiguous {self.dtype} tensor of shape {shape} on {self.device}")
        self._check_buf(done, 0, "done", torch.uint8)
        self._ensure_prepared(weights, "actor")
        self._bind()
        L.check(L.lib.qenv_policy_rollout(self._h, int(W[0].shape[1]), int(W[0].shape[0]), *[L.ptr(t) for t in W], n_steps,
                                          L.ptr(actions), L.ptr(reward), L.ptr(done), L.ptr(mask)), "qenv_policy_rollout")

    def set_ppo_fused(self, on=True, block=None):
        """Whether PolicyRollout.run (in-band) and policy_rollout use the ONE fused launch (policy + env step of the whole
        call, k_ppo_fused; the default) where it applies, or the separate policy / env-step launches.  Bit-identical results.
        block: force 32- or 64-env blocks per wavefront (default: 32 for batches that would leave half the wavefronts idle);
        block=None leaves a block forced earlier as it is (set_ppo_fused_block(None) clears it)."""
        self._bind()
        if block not in (None, 32, 64):
            raise ValueError("block: None (chosen by the batch size), 32 or 64 envs per wavefront")
        L.check(L.lib.qenv_set_ppo_fused(self._h, ({32: 5, 64: 6}[block] if (on and block) else int(bool(on)))), "qenv_set_ppo_fused")

    @property
    def ppo_fused(self):
        """(on, block): whether the fused rollout launch is enabled on this handle and the forced env block (None: by size)."""
        v = int(L.lib.qenv_get_ppo_fused(self._h))
        return bool(v & 1), ((v >> 8) or None)

    def set_ppo_fused_block(self, block=None):
        """Only the env block of the fused launch (None = chosen by the batch size, 32 or 64 envs per wavefront); whether the
        fused launch is used at all stays as set_ppo_fused / QENV_PPO_FUSED left it.  No stream is bound here."""
        if block not in (None, 32, 64):
            raise ValueError("block: None (chosen by the batch size), 32 or 64 envs per wavefront")
        L.check(L.lib.qenv_set_ppo_fused_block(self._h, int(block or 0)), "qenv_set_ppo_fused_block")

    def ppo_flush(self):
        """Complete the warm-up a ppo_step left pending (no-op otherwise); call before reading the env's fields."""
        self._bind()
        L.check(L.lib.qenv_ppo_flush(self._h), "qenv_ppo_flush")

    def sample_action(self, mean, std, noise=None, seed=0, step=0, env_offset=0, out=None, logprob=None):
        """ActorCritic.act sampling (model.py:62-68): mean + std*eps and the per-dimension log-prob.
        noise: injected standard-normal [4, N] (for CPU/GPU parity runs); otherwise Philox keyed by
        (seed; global env index env_offset+e, step)."""
        out = torch.empty(4, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        nz = None if noise is None else self._as_rows(noise, 4, "noise")
        for t, nm in ((mean, "mean"), (out, "out"), (logprob, "logprob")):
            self._check_buf(t, 4, nm)
        self._bind()
        L.check(L.lib.qenv_sample_action(self._h, L.ptr(mean), float(std), L.ptr(nz), int(seed), int(step),
                                         int(env_offset), L.ptr(out), L.ptr(logprob)), "qenv_sample_action")
        return out

    # ---- outer position loops around the velocity policy (position.py:233-244, math_trajectory.py:118-133) ----
    outer_state = property(lambda s: s._field(L.FIELD_OUTER))       # [9, N] integral, previous error, command
    vel_cmd = property(lambda s: s._field(L.FIELD_OUTER)[6:9])      # [3, N] velocity command of the last step

    def outer_velpid(self, table):
        """vel_pid (velocity_pid.py:16-35) on a mission table [n, 7] (mission.table()); None switches it off.
        From now on every step(push_window=True) pushes state - (target + pid_error) into the policy window."""
        if table is not None:
            if isinstance(table, np.ndarray):
                table = torch.from_numpy(np.ascontiguousarray(table))
            table = table.to(device=self.device, dtype=self.dtype).contiguous()
            if table.dim() != 2 or table.shape[1] != 7:
                raise ValueError(f"mission table: expected [n, 7], got {tuple(table.shape)}")
        self._outer_target = table
        self._outer_kind = None if table is None else "velpid"
        L.check(L.lib.qenv_outer_velpid(self._h, L.ptr(table), 0 if table is None else int(table.shape[0])),
                "qenv_outer_velpid")

    def outer_pospi(self, targets, P=4.5, D=0.5, I=0.0):
        """pos_pi + velocity clip (math_trajectory.py:46-62,:123) towards per-env targets [3, N]; None = off."""
        t = self._as_rows(targets, 3, "targets")
        self._outer_target = t
        self._outer_kind = None if t is None else "pospi"
        L.check(L.lib.qenv_outer_pospi(self._h, L.ptr(t), float(P), float(D), float(I)), "qenv_outer_pospi")

    def outer_reset(self, mask=None):
        m = self._mask(mask)
        self._bind()
        L.check(L.lib.qenv_outer_reset(self._h, L.ptr(m)), "qenv_outer_reset")

    # ---- robust_control (quadrotor_env.py:84-109): episode plant perturbation + gusts -----------------
    @property
    def robust_control(self):
        return getattr(self, "_robust", False)

    @robust_control.setter
    def robust_control(self, on):
        if on and self.handle == "lean":   # (qenv_set_robust says the same; here with the Python spelling of the remedy)
            raise ValueError("robust_control: the lean fp64 handle (BatchedQuad's default) has no robust kernels -- create the "
                             "env with dtype=torch.float32, dtype=torch.float64 (IEEE) or dtype=torch.float64, compute='f32'")
        L.check(L.lib.qenv_set_robust(self._h, int(bool(on))), "qenv_set_robust")
        self._robust = bool(on)

    def robust_params(self, params=None, seed=0, env_offset=0, mask=None):
        """robust_control.reset for the masked envs: params [12, N] (reference draw order: kf4, m, ir4, J3) or
        None = the same distributions from Philox keyed by (seed; global env, episode)."""
        p = self._as_rows(params, 12, "params")
        m = self._mask(mask)
        self._bind()
        L.check(L.lib.qenv_robust_params(self._h, L.ptr(p), int(seed), int(env_offset), L.ptr(m)), "qenv_robust_params")

    def robust_gusts(self, table=None, seed=0, env_offset=0):
        """Source of the gust draws: table [E, K, 3, N] (recorded draws, parity runs; kept alive here) or None =
        Philox keyed by (seed; global env, episode, epoch, draw)."""
        if table is not None:
            if isinstance(table, np.ndarray):
                table = torch.from_numpy(np.ascontiguousarray(table))
            table = table.to(device=self.device, dtype=self.dtype).contiguous()
            if table.dim() != 4 or table.shape[2] != 3 or table.shape[3] != self.n_envs:
                raise ValueError(f"gust table: expected [E, K, 3, {self.n_envs}], got {tuple(table.shape)}")
            E, K = int(table.shape[0]), int(table.shape[1])
        else:
            E = K = 1
        self._gust_table = table
        L.check(L.lib.qenv_robust_gusts(self._h, L.ptr(table), E, K, int(seed), int(env_offset)), "qenv_robust_gusts")

    def sample_init_states(self, seed, episode=0, env_offset=0, mask=None, out=None):
        """Device-side random branch of quad.reset (same distributions, Philox streams)."""
        out = torch.zeros(13, self.n_envs, device=self.device, dtype=self.dtype) if out is None else out
        m = self._mask(mask)
        self._check_buf(out, 13, "out")
        self._bind()
        L.check(L.lib.qenv_sample_init_states(self._h, int(seed), int(episode), int(env_offset), L.ptr(m), L.ptr(out)),
                "qenv_sample_init_states")
        return out


class _RobustParameters:
    """The attribute bag of the reference's robust_control object (quadrotor_env.py:84-101), filled by quad.reset."""
    D_KF, D_KM, D_M, D_IR = 0.1, 0.1, 0.3, 0.1
    gust_std, gust_period = [[5], [5], [2]], 500

    def __init__(self):
        self.D_J = np.ones(3) * 0.1
        self.episode_kf, self.episode_m, self.episode_ir = np.zeros(4), np.zeros(1), np.zeros(4)
        self.episode_J = np.zeros((3, 3))


class quad:
    """Drop-in for the reference's single-environment class (quadrotor_env.py:111-577 /
    quadrotor_env_opt.py:78-484 with ``variant='opt'``).  numpy in, numpy out, same shapes;
    every step is one HIP launch on a batch of one plus a single small device-to-host copy.
    ``dtype`` defaults to float64 so trajectories match the fp64 reference to ~1e-12."""

    def __init__(self, t_step, n, training=True, euler=0, direct_control=1, T=1, clipped=True,
                 variant="main", integrator="rk45", device="cuda", dtype=torch.float64):
        self._b = BatchedQuad(1, t_step, n, training=training, euler=euler, direct_control=direct_control, T=T,
                              clipped=clipped, variant=variant, integrator=integrator, device=device, dtype=dtype)
        b = self._b
        self.clipped = clipped
        self.ppo_training = bool(training)
        self.mass, self.gravity = M, G
        self.T = T
        self.bb_cond = b.bb_cond
        self.state_size, self.action_size = 13, 4
        self.n = b.n
        self.t_step = t_step
        self.zero_control = b.zero_control
        self.direct_control_flag = direct_control
        self.J_mat = J
        self.robust_parameters = _RobustParameters()
        if variant == "main":  # robust_control.__init__ calls its own reset() (:85-91): the constructor consumes 12 draws
            params = draw_robust_params(np.random)
            rp = self.robust_parameters
            rp.episode_kf, rp.episode_m, rp.episode_ir = params[0:4], params[4:5], params[5:9]
            rp.episode_J = np.eye(3) * params[9:12]
        self.i = 0
        self.done = True
        self.solved = 0
        self.reward = 0
        self.abs_sum = 0
        self.ang = np.zeros(3)
        self.ang_vel = np.zeros(3)
        self.prev_ang = np.zeros(3)
        self.action_hist = []
        self.target_state = 9 * ((0.001 if variant == "opt" else 0.005) ** 2)
        ev_cd = 'Training' if self.ppo_training else 'Eval'
        print('Environment Condition: ' + ev_cd)  # quadrotor_env.py:185-187

    def seed(self, seed):
        np.random.seed(seed)  # the reference seeds the GLOBAL numpy stream (:193)

    # the reference hard-codes robust_control = False (:183); users flip the attribute
    @property
    def robust_control(self):
        return self._b.robust_control

    @robust_control.setter
    def robust_control(self, on):
        self._b.robust_control = on

    _GUST_K = 6002  # gust triples staged per redraw step (1000 attempted sub-steps; the reference's runs use 8..300)

    def _stage_gusts(self):
        """A step with i % 500 == 1 draws one gust triple per RHS evaluation from the GLOBAL numpy stream, a data
        dependent number of times.  Stage the next _GUST_K triples from a snapshot of the stream as the device's
        table; after the step the stream is advanced by exactly the draws the integrator consumed."""
        self._rng_snapshot = np.random.get_state()
        z = np.random.normal(0.0, 1.0, self._GUST_K * 3).reshape(self._GUST_K, 3) * np.array([5.0, 5.0, 2.0])
        self._b.robust_gusts(torch.from_numpy(z.reshape(1, self._GUST_K, 3, 1).copy()))

    def _commit_gusts(self, n_steps_with_redraw):
        used = int(self._b.gust_evals[0].item())
        if used >= self._GUST_K:
            raise RuntimeError(f"robust_control: the redraw step consumed {used} gust draws, more than the {self._GUST_K} staged")
        np.random.set_state(self._rng_snapshot)
        if used:
            np.random.normal(0.0, 1.0, used * 3)

    def _pull(self):
        """One D2H copy of the whole (tiny) workspace, then refresh the numpy attributes."""
        b = self._b
        host = b._ws.cpu()
        base = b._ws.data_ptr()

        def rows(fid, dt=None):
            t = b._field(fid, dt)
            off = t.data_ptr() - base
            dt = dt or b.dtype
            return host[off:off + t.numel() * t.element_size()].view(dt).numpy().astype(
                np.float64 if dt != torch.int32 else np.int64).reshape(t.shape[0])

        self.state = rows(L.FIELD_STATE)
        self.previous_state = self.state
        self.ang = rows(L.FIELD_ANG)
        self.prev_ang = rows(L.FIELD_PREV_ANG)
        self.ang_vel = rows(L.FIELD_ANG_VEL)
        self.V_q = rows(L.FIELD_VQ)
        self.w = rows(L.FIELD_W).reshape(4, 1)
        self.step_effort = rows(L.FIELD_STEP_EFFORT)
        self.clipped_action = rows(L.FIELD_CLIPPED_ACTION)
        self.accel = rows(L.FIELD_ACCEL).reshape(3, 1)
        self.reward = float(rows(L.FIELD_REWARD)[0])
        self.abs_sum = float(rows(L.FIELD_ABS_SUM)[0])
        self.i = int(rows(L.FIELD_I, torch.int32)[0])
        fl = int(rows(L.FIELD_FLAGS, torch.int32)[0])
        self.done = bool(fl & L.FLAG_DONE)
        self.solved = 1 if fl & L.FLAG_SOLVED else 0
        self.prev_shaping = float(rows(L.FIELD_PREV_SHAPING)[0]) if fl & L.FLAG_HAS_SHAPING else None
        self.quat_state = np.array([np.concatenate((self.state[0:10], self.V_q))])
        self.current_state = float(np.sum(np.square(np.concatenate((self.state[1:6:2], self.ang, self.state[-3:])))))

    # lazily derived side outputs of the last RHS call (drone_eq :315,:352,:371)
    @property
    def mat_rot(self):
        q = self.state[6:10] / np.linalg.norm(self.state[6:10])
        a, b, c, d = q
        return np.array([[a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
                         [2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b],
                         [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d]])

    @property
    def f_in(self):
        return np.array([[0, 0, self.clipped_action[0] if not self.direct_control_flag
                          else float(np.sum((self.clipped_action + 1) * T2WR * M * G / 8))]]).T

    @property
    def accelerometer_read(self):
        return self.mat_rot.T @ (self.accel.flatten() + np.array([0, 0, -G]))

    def reset(self, det_state=None):
        b = self._b
        params = None
        if b.variant == "main":  # robust_parameters.reset() draws first, whatever det_state is (:426)
            params = draw_robust_params(np.random)
            rp = self.robust_parameters
            rp.episode_kf, rp.episode_m, rp.episode_ir = params[0:4], params[4:5], params[5:9]
            rp.episode_J = np.eye(3) * params[9:12]
        if det_state is None:
            ang = np.random.rand(3) - 0.5  # reference draw order (:440-445)
            x0 = np.zeros((1, 13))
            x0[0, 0:5:2] = np.clip(np.random.normal([0, 0, 0], 2), -BB_POS / 2, BB_POS / 2)
            x0[0, 1:6:2] = np.clip(np.random.normal([0, 0, 0], 2), -BB_VEL / 2, BB_VEL / 2)
            x0[0, 6:10] = euler_quat(ang)
            x0[0, 10:13] = np.clip(np.random.normal([0, 0, 0], 2), -BB_VEL * 1.5, BB_POS * 1.5)
        else:
            x0 = np.asarray(det_state, dtype=np.float64).reshape(1, 13)
        self.action_hist = [self.zero_control for _ in range(self.T)]
        robust = self.robust_control
        if robust:
            b.robust_params(torch.from_numpy(params.reshape(12, 1).copy()))
            self._stage_gusts()  # the first warm-up step has i == 1: a redraw step
        states, actions = b.reset(torch.from_numpy(x0.T.copy()))
        if robust:
            self._commit_gusts(1)
        self.action = self.zero_control
        self._pull()
        return (states[:, :, 0].double().cpu().numpy(), actions[:, :, 0].double().cpu().numpy())

    def step(self, action):
        b = self._b
        a = np.asarray(action, dtype=np.float64).reshape(4)
        self.action = np.clip(a, -1, 1) if self.direct_control_flag else a
        redraw = self.robust_control and (self.i + 1) % 500 == 1
        if redraw:
            self._stage_gusts()
        b.step(torch.from_numpy(a.reshape(4, 1)))
        if redraw:
            self._commit_gusts(1)
        self._pull()
        self.action_hist.append(self.clipped_action)
        return self.quat_state, self.reward, self.done
