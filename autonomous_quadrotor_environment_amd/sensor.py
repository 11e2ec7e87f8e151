"""Batched onboard-sensor model: class `sensor` of the reference (environment/quadrotor_env.py:579-724) for N envs,
evaluated as visual_landing/rl_worker.py:164-175 (`sensor_sp`) does after every quad.step: IMU with bias drift and
white noise, GPS, TRIAD attitude, dead-reckoning integrators.  Noise comes from Philox streams keyed by (seed; global env
index, step) or from injected standard normals (the reference's draw order), so runs are reproducible and
shard-invariant."""
import torch

from . import _lib as L
from .quadrotor_env import quad


class BatchedSensor:
    def __init__(self, env, seed=0, env_offset=0, error=True):
        self.env = env._b if isinstance(env, quad) else env
        self.seed, self.env_offset, self.error = int(seed), int(env_offset), bool(error)
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
                                        int(self.error)), "qenv_sensor_reset")
        self.episode += 1

    def sensor_sp(self, z27=None, out=None):
        """rl_worker.py:164-175.  Returns [24, N]: states_sens[14] (x, vx, y, vy, z, vz dead-reckoned; quaternion_gyro;
        quaternion_vel), gps_pos[3], gps_vel[3], quaternion_triad[4].  The env must have been stepped with the default
        (FULL) flags.  z27: injected [27, N] standard normals in the reference's draw order."""
        b = self.env
        out = torch.empty(24, b.n_envs, device=b.device, dtype=b.dtype) if out is None else out
        z = None if z27 is None else b._as_rows(z27, 27, "z27")
        b._bind()
        L.check(L.lib.qenv_sensor_read(b._h, L.ptr(self._ws), L.ptr(z), self.seed, self.calls, self.env_offset,
                                       int(self.error), L.ptr(out)), "qenv_sensor_read")
        self.calls += 1
        return out
