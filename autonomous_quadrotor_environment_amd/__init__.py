"""MI355X-native batched quadrotor RL environment.

Drop-in for the hot path of rafaelcostafrf/autonomous_quadrotor_environment: the `quad`
reset()/step()/reward API (environment/quadrotor_env.py), the LQR / PID inner-loop
controllers, the PPO observation window + actor MLP and GAE -- executed by hand-written
HIP kernels for gfx950 behind the C-ABI in include/qenv.h.  PyTorch is used only for
device memory, streams and torch.distributed.
"""
from . import _lib  # noqa: F401  (raises if libqenv.so is missing: no CPU fallback)
from .quadrotor_env import BatchedQuad, quad, random_init_states, euler_quat, quat_euler  # noqa: F401
from .controllers import LqrController, PidController, PidPosController, pid_control  # noqa: F401
from .ppo import (ActorCriticWeights, dense_tail, dl_in_gen, gae, adv_normalize, ppo_loss, visual_reward, PolicyRollout,  # noqa: F401
                  StreamShardedRollout, PpoLearner)
from .mission import mission  # noqa: F401
from .sensor import BatchedSensor, sensor  # noqa: F401
from .position_control import MissionTracker, LandingBaseline  # noqa: F401
from .distributed import shard_range, ShardedQuad, all_gather_returns, GradBucket  # noqa: F401

__all__ = ["BatchedQuad", "quad", "random_init_states", "LqrController", "PidController", "PidPosController", "pid_control",
           "ActorCriticWeights", "dl_in_gen", "gae", "dense_tail", "ppo_loss", "visual_reward", "PolicyRollout", "StreamShardedRollout", "adv_normalize", "PpoLearner", "mission", "BatchedSensor", "sensor", "MissionTracker", "LandingBaseline", "shard_range",
           "ShardedQuad", "all_gather_returns", "GradBucket"]
