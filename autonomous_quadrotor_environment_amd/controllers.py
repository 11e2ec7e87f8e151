"""Inner-loop controllers of the reference, evaluated by HIP kernels.

* LQR  : environment/controller/lqr_quad.py:25-111 (gains, closed-form CARE solution of the
         decoupled double integrators) and :128-154 (law)
* PID  : environment/controller/pid_vel_control.py:29-127
Both read the env's state / Euler angles / finite-difference Euler rates from the workspace
and write the [F, Mx, My, Mz] request; `BatchedQuad.rollout('lqr'|'pid', n)` fuses them with
the env step in one launch.
"""

from .quadrotor_env import quad


def _batched(env):
    return env._b if isinstance(env, quad) else env


class LqrController:
    """state -> action = [U1, Mx, My, Mz] for every env (lqr_quad.py:128-154)."""

    def __init__(self, env):
        self.env = env
        self._b = _batched(env)
        if self._b.direct_control_flag:
            raise ValueError("LQR commands [F, M]: create the env with direct_control=0 (lqr_quad.py:116)")

    def control(self, out=None):
        a = self._b.lqr_action(out)
        return a[:, 0].double().cpu().numpy() if isinstance(self.env, quad) else a


class PidController:
    """Cascaded velocity -> attitude PID (pid_vel_control.py:29-110), per-env state on device."""

    def __init__(self, env):
        self.env = env
        self._b = _batched(env)
        if self._b.direct_control_flag:
            raise ValueError("PID commands [F, M]: create the env with direct_control=0 (pid_vel_control.py:131)")
        self._b.pid_reset()  # a fresh pid_control per episode (pid_vel_control.py:144)

    def reset(self, mask=None):
        self._b.pid_reset(mask)

    def control(self, xd=(0, 0, 0), dxd=(0, 0, 0), psd=0, dpsd=0, out=None):
        a = self._b.pid_action([xd[0], xd[1], xd[2], psd], out)
        return a[:, 0].double().cpu().numpy() if isinstance(self.env, quad) else a


class pid_control(PidController):
    """Name-compatible with the reference class (pid_vel_control.py:29)."""

    def __init__(self, drone_env):
        super().__init__(drone_env)
        self.att_target = []


class PidPosController(PidController):
    """Position-tracking PD of environment/controller/PID_control.py:29-106 (trajectory targets)."""

    def control(self, xd=(0, 0, 0), dxd=(0, 0, 0), psd=0, dpsd=0, out=None):
        a = self._b.pidpos_action([xd[0], xd[1], xd[2], dxd[0], dxd[1], dxd[2], psd], out)
        return a[:, 0].double().cpu().numpy() if isinstance(self.env, quad) else a
