"""Reference trajectories for the position-tracking loop (mission_control/mission_control.py:3-83), as dense
tables the device rollout can walk: `table(n)` returns the [n, 7] array {xd, yd, zd, dxd, dyd, dzd, psi_d} that
`mission.get_error` would hand out call after call (PID_control.py:250-263), ready for
BatchedQuad.rollout('pidpos', n, targets=table).  Host-side array code, as in the reference."""
import numpy as np


class mission:
    def __init__(self, time_step):
        self.time_step = time_step
        self.trajectory = np.zeros((0, 3))
        self.velocity = np.zeros((0, 3))
        self.trajectory_step = 0
        self.trajectory_total_steps = 0

    def _finite_difference_velocity(self, steps):
        # velocity[i+1] = (trajectory[i+1] - trajectory[i]) / dt for i < steps-1 (mission_control.py:23-26)
        if steps > 1:
            self.velocity[1:steps] = (self.trajectory[1:steps] - self.trajectory[0:steps - 1]) / self.time_step

    def gen_trajectory(self, total_timesteps, steps, position, velocity=None, additive=None):
        """Ramp to `position` in `steps` steps then hold (velocity=None), or constant-acceleration ramp to
        `velocity` (mission_control.py:7-31)."""
        self.trajectory_step, self.trajectory_total_steps = 0, steps
        origin = np.zeros(14) if additive is None else np.asarray(additive, dtype=float)
        self.trajectory = np.zeros([total_timesteps, 3])
        self.velocity = np.zeros([total_timesteps, 3])
        if velocity is None:
            for i in range(3):
                self.trajectory[:steps, i] = np.linspace(origin[i], position[i] + origin[i], steps)
                self.trajectory[steps:, i] = position[i]
            self._finite_difference_velocity(steps)
        else:
            for i in range(3):
                self.velocity[:, i] = np.linspace(0, velocity[i], steps)
            if steps > 1:
                self.trajectory[1:steps] = np.cumsum(self.velocity[0:steps - 1] * self.time_step, axis=0)

    def sin_trajectory(self, steps, circular_rate, ascent_rate, center, axis):
        """mission_control.py:33-45 (z is a running sum that starts from the LAST row, as the reference's
        trajectory[step-1] with step = 0 reads)."""
        self.trajectory_step, self.trajectory_total_steps = 0, steps
        self.trajectory = np.zeros([steps, 3])
        self.velocity = np.zeros([steps, 3])
        a = np.arange(steps) * circular_rate * self.time_step
        self.trajectory[:, :] = np.asarray(center, dtype=float) + np.sin(a)[:, None] * np.asarray(axis, dtype=float)
        z = 0.0  # trajectory[-1, 2] is still zero when step 0 reads it
        for step in range(steps):
            z = z + ascent_rate * self.time_step
            self.trajectory[step, 2] = z
        self._finite_difference_velocity(steps)

    def spiral_trajectory(self, zsteps, steps, rate, circular_rate, radius, center):
        """mission_control.py:48-64."""
        self.trajectory_step, self.trajectory_total_steps = 0, steps
        self.trajectory = np.zeros([steps, 3])
        self.velocity = np.zeros([steps, 3])
        k = np.arange(steps)
        a = k * circular_rate * self.time_step
        z = np.where(k > zsteps, zsteps * rate * self.time_step, k * rate * self.time_step)
        self.trajectory[:, :] = np.asarray(center, dtype=float) + np.stack([np.cos(a) * radius, np.sin(a) * radius, z], 1) \
            - np.array([radius, 0, 0])
        self._finite_difference_velocity(steps)

    def get_error(self, time=None):
        """Next target as the reference's 14-vector [x, vx, y, vy, z, vz, 0...] (mission_control.py:67-83),
        including its end-of-trajectory behaviour (the last point keeps drifting with the last velocity)."""
        if self.trajectory_step == self.trajectory_total_steps:
            self.trajectory[-1, :] = self.trajectory[-1, :] + self.velocity[-1, :] * self.time_step
            p, v = self.trajectory[-1], self.velocity[-1]
        else:
            p, v = self.trajectory[self.trajectory_step], self.velocity[self.trajectory_step]
            self.trajectory_step += 1
        out = np.zeros(14)
        out[0:5:2] = p
        out[1:6:2] = v
        return out

    def table(self, n, psi_d=0.0):
        """The next n get_error() results as a [n, 7] array {xd, yd, zd, dxd, dyd, dzd, psi_d}."""
        out = np.zeros((n, 7))
        for k in range(n):
            e = self.get_error()
            out[k, 0:3], out[k, 3:6], out[k, 6] = e[0:5:2], e[1:6:2], psi_d
        return out
