"""ControlEnv — model x task factory with the reference's signature (envs/control_env.py:12-35)."""
from .env_base import BaseEnv
from .models.F16_model import F16Model
from .tasks.task_base import ControlTask, HeadingTask, TrackingTask

_TASKS = {'heading': HeadingTask, 'control': ControlTask, 'tracking': TrackingTask}


class ControlEnv(BaseEnv):
    """Fly-control env: one F-16 per agent, tasks heading / control / tracking."""

    def __init__(self, num_envs=1, config='heading', model='F16', random_seed=None, device='cuda:0', row0=0,
                 aero_1d_tables=None, solver=None, weights=None, airframe=None):
        super().__init__(num_envs, config, model, random_seed, device, row0=row0, aero_1d_tables=aero_1d_tables, solver=solver,
                         weights=weights, airframe=airframe)

    def load(self, random_seed, config, model):
        if model != 'F16':
            # the reference's only other model (UAV) crashes on its 2nd reset with the shipped
            # YAMLs; it is out of the accelerated path
            raise NotImplementedError
        if config not in _TASKS:
            raise NotImplementedError
        batch = self._make_batch(config, random_seed)
        self.model = F16Model(self.config, self.n, self.device, random_seed, batch)
        self.task = _TASKS[config](self.config, self.n, self.device, random_seed, batch)
        if random_seed is not None:      # env_base.py:43-44: construction with a seed seeds the process-wide generators too
            self.model.seed(random_seed)
