"""Onboard-sensor model: class `sensor` of the reference (environment/quadrotor_env.py:579-724).

* ``BatchedSensor`` -- N envs, evaluated as visual_landing/rl_worker.py:164-175 (`sensor_sp`) does after every quad.step:
  IMU with bias drift and white noise, GPS, TRIAD attitude, dead-reckoning integrators, one launch per call; or fused
  into the env step (``attach()``: the policy window is fed from the sensed state, position.py:240-242).
* ``sensor`` -- the reference's class for ONE env: same constructor, attributes and methods (``accel``, ``gyro``, ``gps``,
  ``triad``, ``accel_int``, ``gyro_int``, ``reset``, ``bias_reset``), numpy in / numpy out, so rl_worker.py-style code runs
  unchanged on the `quad` drop-in.  Every number comes from the device kernels; the noise is drawn on the host from the
  global numpy stream exactly where the reference draws it, so seeded scripts reproduce.

Batched noise comes from Philox streams keyed by (seed; global env index, step) or from injected standard normals (the
reference's draw order), so runs are reproducible and shard-invariant."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .quadrotor_env import quad

ACCEL, GYRO, GPS, TRIAD, ACCEL_INT, GYRO_INT = range(6)
_N_NORMALS = {ACCEL: 3, GYRO: 3, GPS: 6, TRIAD: 6, ACCEL_INT: 9, GYRO_INT: 3}
_N_OUT = {ACCEL: 3, GYRO: 3, GPS: 6, TRIAD: 13, ACCEL_INT: 9, GYRO_INT: 4}
_DEFAULTS = (0.1, 0.035, 15, 1.71, 0.5, 0.0005, 0.00015, 0.075)  # std: accel gyro magnet gps_p gps_v | drift: accel gyro magnet


def _params(p):
    return None if p is None else (C.c_double * 8)(*[float(x) for x in p])


class BatchedSensor:
    def __init__(self, env, seed=0, env_offset=0, error=True, params=None):
        """params: None = the reference's defaults, or the eight values of sensor.__init__ (:588-592):
        accel_std, gyro_std, magnet_std, gps_std_p, gps_std_v, accel_bias_drift, gyro_bias_drift, magnet_bias_drift."""
        self.env = env._b if isinstance(env, quad) else env
        self.seed, self.env_offset, self.error = int(seed), int(env_offset), bool(error)
        self.params = None if params is None else tuple(float(x) for x in params)
        b = self.env
        self._ws = torch.zeros(L.SENSOR_ROWS, b.n_envs, device=b.device, dtype=b.dtype)
        self._ws[15] = 1; self._ws[19] = 1; self._ws[23] = 1  # self.R = np.eye(3) (:599)
        self.calls = 0
        self.episode = 0

    def reset(self, u3=None):
        """sensor.reset() (:621-631) + bias_reset (:601-609).  u3: injected [3, N] uniforms (else Philox)."""
        b = self.env
        u = None if u3 is None else b._as_rows(u3, 3, "u3")
        b._bind()
        L.check(L.lib.qenv_sensor_reset(b._h, L.ptr(self._ws), L.ptr(u), self.seed, self.episode, self.env_offset,
                                        int(self.error), _params(self.params)), "qenv_sensor_reset")
        self.episode += 1

    def bias_reset(self, u3):
        """bias_reset (:601-609) alone from three uniforms per env [3, N]."""
        b = self.env
        b._bind()
        L.check(L.lib.qenv_sensor_bias_reset(b._h, L.ptr(self._ws), L.ptr(b._as_rows(u3, 3, "u3")), int(self.error),
                                             _params(self.params)), "qenv_sensor_bias_reset")

    def sensor_sp(self, z27=None, out=None):
        """rl_worker.py:164-175.  Returns [24, N]: states_sens[14] (x, vx, y, vy, z, vz dead-reckoned; quaternion_gyro;
        quaternion_vel), gps_pos[3], gps_vel[3], quaternion_triad[4].  The env must have been stepped with the default
        (FULL) flags.  z27: injected [27, N] standard normals in the reference's draw order."""
        b = self.env
        out = torch.empty(24, b.n_envs, device=b.device, dtype=b.dtype) if out is None else out
        b._check_buf(out, 24, "out")
        z = None if z27 is None else b._as_rows(z27, 27, "z27")
        b._bind()
        L.check(L.lib.qenv_sensor_read(b._h, L.ptr(self._ws), L.ptr(z), self.seed, self.calls, self.env_offset,
                                       int(self.error), _params(self.params), L.ptr(out)), "qenv_sensor_read")
        self.calls += 1
        return out

    def ops(self, ops, z):
        """The class's methods in any order (ids ACCEL .. GYRO_INT): z [sum of their normals, N] standard normals in call
        order; returns [sum of their outputs, N] (accel 3, gyro 3, gps pos 3 + vel 3, triad q 4 + R 9, accel_int
        acceleration 3 + velocity 3 + position 3, gyro_int q 4)."""
        b = self.env
        ops = [int(o) for o in ops]
        nz, no = sum(_N_NORMALS[o] for o in ops), sum(_N_OUT[o] for o in ops)
        zt = b._as_rows(z, nz, "z")
        out = torch.empty(no, b.n_envs, device=b.device, dtype=b.dtype)
        arr = (C.c_int32 * len(ops))(*ops)
        b._bind()
        L.check(L.lib.qenv_sensor_ops(b._h, L.ptr(self._ws), arr, len(ops), L.ptr(zt), int(self.error), _params(self.params),
                                      L.ptr(out)), "qenv_sensor_ops")
        return out

    def attach(self, z27=None, out=None):
        """Fuse sensor_sp into the env's PUSH_WINDOW steps (position.py:240-242, REAL_CTRL = False): from now on
        step(push_window=True) evaluates this sensor on the state the step ended in, inside the step kernel, and the
        policy window receives [action, dead-reckoned velocity - command, gyro-integrated quaternion, its rate] instead
        of the true state.  z27: the [27, N] tensor the caller refills with injected normals before every step (None:
        Philox keyed by (seed; env, sensor call)); out: optional [24, N] tensor receiving sensor_sp's outputs."""
        b = self.env
        if z27 is not None and (tuple(z27.shape) != (27, b.n_envs) or z27.dtype != b.dtype or not z27.is_contiguous()):
            raise ValueError(f"z27: a contiguous [27, {b.n_envs}] {b.dtype} tensor the caller refills in place")
        b._check_buf(out, 24, "out")
        self._attached = (z27, out)
        zt, ot = self._attached
        b._sensor = self
        L.check(L.lib.qenv_attach_sensor(b._h, L.ptr(self._ws), L.ptr(zt), self.seed, self.env_offset, int(self.error),
                                         _params(self.params), L.ptr(ot), self.calls), "qenv_attach_sensor")
        return self

    def detach(self):
        L.check(L.lib.qenv_attach_sensor(self.env._h, None, None, 0, 0, 0, None, None, 0), "qenv_attach_sensor")
        self._attached = None
        self.env._sensor = None


class sensor:
    """Drop-in for the reference's `sensor(env, ...)` on the single-env `quad` (quadrotor_env.py:579-724)."""

    def __init__(self, env, accel_std=0.1, accel_bias_drift=0.0005, gyro_std=0.035, gyro_bias_drift=0.00015,
                 magnet_std=15, magnet_bias_drift=0.075, gps_std_p=1.71, gps_std_v=0.5):
        if not isinstance(env, quad):
            raise TypeError("sensor(env): env must be the single-env `quad` (use BatchedSensor for BatchedQuad)")
        self.std = [accel_std, gyro_std, magnet_std, gps_std_p, gps_std_v]
        self.b_d = [accel_bias_drift, gyro_bias_drift, magnet_bias_drift]
        self.quad = env
        self._error = True
        self._s = BatchedSensor(env, error=True, params=self.std + self.b_d)
        self.bias_reset()
        self.R = np.eye(3)
        # reset() defines these in the reference; reading them earlier raises there too
        self.a_b_accel = self.g_b = self.a_b_grav = self.m_b = 0

    # `error` is a plain attribute the reference's users flip (sens.error = False) before reset()
    @property
    def error(self):
        return self._error

    @error.setter
    def error(self, on):  # takes effect at the next bias_reset() / reset(), where the reference folds it into a_std ... (:601-609)
        self._error = bool(on)

    def _row(self, r, n=1):
        return self._s._ws[r:r + n, 0].double().cpu().numpy()

    def _sync_attrs(self):
        w = self._s._ws[:, 0].double().cpu().numpy()
        self.a_b_d, self.g_b_d, self.m_b_d = float(w[0]), float(w[1]), float(w[2])
        self.a_b_accel, self.g_b = float(w[3]), float(w[4])
        self.position_t0, self.velocity_t0 = w[5:8].copy(), w[8:11].copy()
        self.quaternion_t0 = w[11:15].copy()
        self.R = w[15:24].reshape(3, 3).copy()

    def bias_reset(self):
        e = float(self._error)
        self.a_std, self.g_std, self.m_std = self.std[0] * e, self.std[1] * e, self.std[2] * e
        self.gps_std_p, self.gps_std_v = self.std[3] * e, self.std[4] * e
        u = np.array([np.random.random(), np.random.random(), np.random.random()])  # the reference's three draws (:603-607)
        self._s.error = self._error
        self._s.bias_reset(torch.from_numpy(u.reshape(3, 1)))
        self._sync_attrs()

    def reset(self):
        u = np.array([np.random.random(), np.random.random(), np.random.random()])
        e = float(self._error)
        self.a_std, self.g_std, self.m_std = self.std[0] * e, self.std[1] * e, self.std[2] * e
        self.gps_std_p, self.gps_std_v = self.std[3] * e, self.std[4] * e
        self._s.error = self._error
        self._s.reset(torch.from_numpy(u.reshape(3, 1)))
        self.a_b_grav = 0
        self.m_b = 0
        self.acceleration_t0 = np.zeros(3)
        self._sync_attrs()

    def _call(self, op):
        # np.random.normal(loc, scale, 3) of the reference = loc + scale * legacy gauss: the same stream positions as
        # standard_normal (the device applies loc and scale)
        z = np.random.standard_normal(_N_NORMALS[op])
        out = self._s.ops([op], torch.from_numpy(z.reshape(-1, 1)))[:, 0].double().cpu().numpy()
        self._sync_attrs()
        return out

    def accel(self):
        return self._call(ACCEL)

    def gyro(self):
        return self._call(GYRO)

    def gps(self):
        o = self._call(GPS)
        return o[0:3], o[3:6]

    def triad(self):
        o = self._call(TRIAD)
        return o[0:4], o[4:13].reshape(3, 3)

    def accel_int(self):
        o = self._call(ACCEL_INT)
        self.acceleration_t0 = o[0:3].copy()
        return o[0:3], o[3:6], o[6:9]

    def gyro_int(self):
        o = self._call(GYRO_INT)
        self.quad._pull()  # the first call after reset() advances quad.state's quaternion in place (:630, :719)
        return o
