"""Task objects of the env surface (reference: envs/tasks/task_base.py and the three task files).

A task here only carries what callers read: spaces, `num_observation/num_actions`, `noise_scale`
and the three per-aircraft target tensors (views into the SoA target buffer).  Target re-draw,
observation, reward and termination are fused into the HIP step/reset kernels
(heading_task.py:49-152, control_task.py:49-152, tracking_task.py:48-155, task_base.py:60-96).
"""
import numpy as np

from ..spaces import Box


class BaseTask:
    target_names = ()

    def __init__(self, config, n, device, random_seed, batch):
        self.config = config
        self.n = n
        self.device = device
        self._b = batch
        self.num_observation = getattr(config, 'num_observation', 12)
        self.num_actions = getattr(config, 'num_actions', 5)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.num_observation,))
        self.action_space = Box(low=-np.inf, high=np.inf, shape=(self.num_actions,))

    @property
    def noise_scale(self):
        return self._b.noise_scale

    # -- the reference's task protocol (task_base.py:33-96) ------------------------------------------------------------------------
    def load_observation_space(self):
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.num_observation,))

    def load_action_space(self):
        self.action_space = Box(low=-np.inf, high=np.inf, shape=(self.num_actions,))

    def seed(self, random_seed):
        """task_base.py:45-49: the process-wide generators, as BaseModel.seed."""
        from ..models.model_base import BaseModel
        BaseModel.seed(self, random_seed)

    def get_obs(self, env):
        """Observation of the env's current state (one kernel launch)."""
        return env.obs()

    def reset(self, env):
        raise RuntimeError('task.reset is fused into BaseEnv.reset()/step() (one HIP kernel)')

    def get_reward(self, env):
        raise RuntimeError('task.get_reward is fused into BaseEnv.step() (one HIP kernel)')

    def get_termination(self, env, info=None):
        raise RuntimeError('task.get_termination is fused into BaseEnv.step() (one HIP kernel)')

    def _target(self, k):
        return self._b.tgt[k]

    def _set_target(self, k, value):
        self._b.tgt[k].copy_(value)


def _target_property(k):
    return property(lambda self: self._target(k), lambda self, v: self._set_target(k, v))


def _members(task, reward_cls, unreach_cls):
    """`reward_functions` / `termination_conditions` as the reference's tasks list them (heading_task.py:34-48 and siblings; Timeout
    is commented out there too).  The objects report what the fused step computed (reward_functions/, termination_conditions/)."""
    from ..reward_functions.event_driven_reward import EventDrivenReward
    from ..termination_conditions.extreme_state import ExtremeState
    from ..termination_conditions.high_speed import HighSpeed
    from ..termination_conditions.low_altitude import LowAltitude
    from ..termination_conditions.low_speed import LowSpeed
    from ..termination_conditions.overload import Overload
    c = task.config
    task.reward_functions = [reward_cls(c), EventDrivenReward(c)]
    task.termination_conditions = [Overload(c), LowAltitude(c), HighSpeed(c), LowSpeed(c), ExtremeState(c), unreach_cls(c, task.device)]


class HeadingTask(BaseTask):
    """targets: altitude [ft], heading [rad], vt [ft/s]  (heading_task.py:26-28)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from ..reward_functions.heading_reward import HeadingReward
        from ..termination_conditions.unreach_heading import UnreachHeading
        _members(self, HeadingReward, UnreachHeading)

    target_altitude = _target_property(0)
    target_heading = _target_property(1)
    target_vt = _target_property(2)


class ControlTask(BaseTask):
    """targets: pitch [rad], heading [rad], vt [ft/s]  (control_task.py:27-29)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from ..reward_functions.posture_reward import PostureReward
        from ..termination_conditions.unreach_posture import UnreachPosture
        _members(self, PostureReward, UnreachPosture)

    target_pitch = _target_property(0)
    target_heading = _target_property(1)
    target_vt = _target_property(2)


class TrackingTask(BaseTask):
    """targets: npos, epos, altitude [ft]  (tracking_task.py:27-29)"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from ..reward_functions.position_reward import PositionReward
        from ..termination_conditions.unreach_target import UnreachTarget
        _members(self, PositionReward, UnreachTarget)

    target_npos = _target_property(0)
    target_epos = _target_property(1)
    target_altitude = _target_property(2)
