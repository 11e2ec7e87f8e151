"""Outer position loops around the PPO velocity policy, for N quadrotors at once, everything on the device:

* MissionTracker  -- environment/position.py:216-244: policy -> step -> mission.get_error -> vel_pid
  (environment/controller/velocity_pid.py:16-35) -> window of `states - error` (REAL_CTRL), or of the SENSED state
  `states_sens - error` (REAL_CTRL = False, :240-242) with a BatchedSensor fused into the step kernel (`sensor=`).
* LandingBaseline -- visual_landing/math_trajectory.py:34-157: random marker / start, PD position loop (pos_pi) ->
  clipped velocity command -> policy, visual_reward (rl_reward_fuction.py) deciding the end of each episode.

Per step: one MFMA launch for the actor and one step launch; the outer loop itself is evaluated inside the step
kernel (qenv_outer_velpid / qenv_outer_pospi), so nothing comes back to the host until the results are read."""
import numpy as np
import torch

from .ppo import ActorCriticWeights, visual_reward
from .quadrotor_env import BatchedQuad

VELOCITY_SCALE = np.array([0.5, 0.5, 1.0])                    # math_trajectory.py:88
VELOCITY_D = np.array([0.0, 0.0, -VELOCITY_SCALE[2] / 1.5])   # :89


class MissionTracker:
    def __init__(self, env: BatchedQuad, weights: ActorCriticWeights):
        if not env.direct_control_flag:
            raise ValueError("the PPO controller commands rotor thrusts: create the env with direct_control=1")
        self.env, self.w = env, weights
        self.action = torch.zeros(4, env.n_envs, device=env.device, dtype=env.dtype)

    def run(self, table, n_steps, init_state=None, log=False, sensor=None):
        """Reset (hover unless init_state is given), then n_steps of policy + outer loop on the mission table
        [n, 7] (mission.table()).  Returns the position error norm summed over the steps (position.py:221,
        `error_sum`) per env, and the [n_steps, 3, N] positions if log.  sensor: a BatchedSensor of this env -- the
        policy then flies on the dead-reckoned / gyro-integrated state (position.py's REAL_CTRL = False), still two
        launches per step."""
        env = self.env
        if init_state is None:
            init_state = np.array([0, 0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0])
        table = np.asarray(table, dtype=np.float64) if not isinstance(table, torch.Tensor) else table
        env.outer_velpid(table)
        env.outer_reset()
        env.reset(init_state, return_history=False, push_window=True)
        if sensor is not None:
            sensor.reset()      # self.sensor.reset() after env.reset (:206)
            sensor.attach()
        tdev = env._outer_target
        err_sum = torch.zeros(env.n_envs, device=env.device, dtype=env.dtype)
        pos_log = torch.empty(n_steps, 3, env.n_envs, device=env.device, dtype=env.dtype) if log else None
        prev_target = torch.zeros(3, 1, device=env.device, dtype=env.dtype)  # error_mission starts as zeros (:198)
        for k in range(n_steps):
            env.mlp_forward(self.w, "actor", out=self.action)
            env.step(self.action, push_window=True)
            pos = env.state[0:5:2]
            err_sum += torch.sqrt(((pos - prev_target) ** 2).sum(0))  # against the PREVIOUS step's target (:221)
            prev_target = tdev[min(k, tdev.shape[0] - 1), 0:3].reshape(3, 1)  # pid_error has no position entries
            if log:
                pos_log[k] = pos
        env.outer_velpid(None)
        if sensor is not None:
            sensor.detach()
        return (err_sum, pos_log) if log else err_sum


class LandingBaseline:
    def __init__(self, env: BatchedQuad, weights: ActorCriticWeights, P=4.5, D=0.5, I=0.0):
        if not env.direct_control_flag:
            raise ValueError("the PPO controller commands rotor thrusts: create the env with direct_control=1")
        self.env, self.w, self.gains = env, weights, (P, D, I)
        self.action = torch.zeros(4, env.n_envs, device=env.device, dtype=env.dtype)

    @staticmethod
    def draw(rng, n):
        """quad_reset_random (math_trajectory.py:34-43), env after env in the script's draw order.
        Returns (marker [n, 3], init_state [n, 13])."""
        if isinstance(rng, (int, np.integer)):
            rng = np.random.RandomState(int(rng))
        marker, x0 = np.zeros((n, 3)), np.zeros((n, 13))
        for k in range(n):
            marker[k] = np.append(rng.normal([0, 0], 0.8), 0.001)
            qz = -5 * rng.random_sample() + 1
            qxy = marker[k, 0:2] + (rng.random_sample(2) - 0.5) * abs(-5 - qz) / 7 * 4
            x0[k] = [qxy[0], 0, qxy[1], 0, qz, 0, 1, 0, 0, 0, 0, 0, 0]
        return marker, x0

    def run(self, marker, init_state, max_steps=3000, log=False):
        """One landing episode per env.  Returns a dict: solved [N] int (n_solved of visual_reward), steps [N]
        (the step at which visual_reward said done; max_steps if never), final_reward [N], delta_v [N]
        (sum of |v| over the episode, :150) and, if log, states [steps, 13, N] / vel_cmd [steps, 3, N]."""
        env = self.env
        N, dev, dt = env.n_envs, env.device, env.dtype
        marker_t = torch.as_tensor(np.asarray(marker).T.copy(), device=dev, dtype=dt)           # [3, N]
        s_d = marker_t.clone()
        s_d[2] = -5.0
        env.outer_pospi(s_d, *self.gains)
        env.outer_reset()
        env.reset(np.asarray(init_state).T.copy(), return_history=False, push_window=True)
        active = torch.ones(N, device=dev, dtype=torch.bool)
        solved = torch.zeros(N, device=dev, dtype=torch.int32)
        steps = torch.full((N,), max_steps, device=dev, dtype=torch.int32)
        final_reward = torch.zeros(N, device=dev, dtype=dt)
        delta_v = torch.zeros(N, device=dev, dtype=dt)
        last_shaping = torch.zeros(N, device=dev, dtype=dt)
        scale = torch.as_tensor(VELOCITY_SCALE, device=dev, dtype=dt).reshape(3, 1)
        vd = torch.as_tensor(VELOCITY_D, device=dev, dtype=dt).reshape(3, 1)
        st_log, cmd_log = [], []
        for k in range(max_steps):
            env.mlp_forward(self.w, "actor", out=self.action)
            env.step(self.action, push_window=True)
            control = (env.vel_cmd - vd) / scale
            kk = torch.full((N,), k, device=dev, dtype=torch.int32)
            rew, sh, done, ns = visual_reward(1 << 30, marker_t, env.state[0:5:2], env.state[1:6:2], control,
                                              last_shaping, kk, env.ang, env.state[10:13])
            last_shaping = torch.where(active, sh, last_shaping)
            delta_v += torch.where(active, env.state[1:6:2].abs().sum(0), torch.zeros_like(delta_v))
            ended = active & done
            solved = torch.where(ended, ns, solved)
            steps = torch.where(ended, kk + 1, steps)
            final_reward = torch.where(ended, rew, final_reward)
            active = active & ~done
            if log:
                st_log.append(env.state.clone()); cmd_log.append(env.vel_cmd.clone())
            if k % 64 == 63 and not bool(active.any()):  # one host sync every 64 steps
                break
        env.outer_pospi(None)
        out = dict(solved=solved, steps=steps, final_reward=final_reward, delta_v=delta_v)
        if log:
            out["states"], out["vel_cmd"] = torch.stack(st_log), torch.stack(cmd_log)
        return out
