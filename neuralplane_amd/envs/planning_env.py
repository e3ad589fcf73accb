"""PlanningEnv — hierarchical tracking env with the reference's surface (envs/planning_env.py:32-177).

One high-level action (Δpitch, Δheading, Δvt) per `step`; inside, 50 low-level iterations of
{low-level observation -> frozen recurrent controller -> fused FDM step}.  Each iteration is two
kernel launches (np_f16_lowlevel_obs, np_f16_step with inner_step=1) plus the controller's forward;
the env keeps the reference's quirks: rows that terminated earlier in the same outer step keep their
state while their controls keep moving, `step_count` advances for every row, flags accumulate.

The controller is whatever the caller passes (`controller(obs, rnn_states, masks, deterministic=True)
-> (actions[n,4], _, rnn_states)`).  With `controller=None` the reference's own `PPOActor` is
imported from the host repository (`algorithms.ppo.ppo_actor`) and loaded from the reference's
checkpoint path — that checkpoint is not shipped with the reference snapshot, so
this raises a clear error unless the host repo provides it.
"""
import os

import numpy as np
import torch

from .env_base import BaseEnv
from .models.F16_model import F16Model
from .spaces import Box
from .tasks.task_base import TrackingTask

INNER_STEPS = 50  # planning_env.py:153


class _ActorArgs:  # planning_env.py:18-29
    def __init__(self, device):
        self.gain = 0.01
        self.hidden_size = '128 128'
        self.act_hidden_size = '128 128'
        self.activation_id = 1
        self.use_feature_normalization = True
        self.use_recurrent_policy = True
        self.recurrent_hidden_size = 128
        self.recurrent_hidden_layers = 1
        self.tpdv = dict(dtype=torch.float32, device=device)
        self.use_prior = False


class PlanningEnv(BaseEnv):
    def __init__(self, num_envs=1, config='tracking', model='F16', random_seed=None, device='cuda:0', controller=None,
                 controller_checkpoint=None, row0=0, aero_1d_tables=None):
        super().__init__(num_envs, config, model, random_seed, device, row0=row0, aero_1d_tables=aero_1d_tables)
        self.low_level_action_space = Box(low=-np.inf, high=np.inf, shape=(4,))
        self.controller = controller if controller is not None else self._load_reference_actor(controller_checkpoint)
        self.ego_rnn_states = torch.zeros((self.n, 1, 128), device=self.device)

    def load(self, random_seed, config, model):
        if model != 'F16':
            raise NotImplementedError
        if config != 'tracking':
            raise NotImplementedError
        batch = self._make_batch(config, random_seed)
        self.model = F16Model(self.config, self.n, self.device, random_seed, batch)
        self.task = TrackingTask(self.config, self.n, self.device, random_seed, batch)

    def _load_reference_actor(self, checkpoint):
        try:
            from algorithms.ppo.ppo_actor import PPOActor  # the host repo's own actor (out of the accelerated path)
        except Exception as e:  # pragma: no cover
            raise RuntimeError('PlanningEnv needs a low-level controller: pass controller=..., or run inside the '
                               'NeuralPlane repo so that algorithms.ppo.ppo_actor.PPOActor is importable') from e
        ckpt = checkpoint or os.path.join(os.getcwd(), '..', 'scripts', 'runs',
                                          '2024-05-26_02-14-24_Control_control_ppo_v1', 'episode_249', 'actor_latest.pt')
        if not os.path.exists(ckpt):
            raise RuntimeError(f'low-level controller checkpoint {ckpt} not found (it is not part of the reference snapshot)')
        actor = PPOActor(_ActorArgs(self.device), self.observation_space, self.low_level_action_space, device=self.device)
        actor.eval()
        actor.load_state_dict(torch.load(ckpt, map_location=self.device))
        return actor

    def low_level_obs(self, target_pitch, target_heading, target_vt):
        """22-float observation of the low-level controller (planning_env.py:60-142)."""
        return self._batch.lowlevel_obs(torch.stack((target_pitch, target_heading, target_vt)))

    def step(self, action, render=False, count=0):
        b = self._batch
        b.reset(want_obs=False)                                    # self.reset()           :145
        action = torch.clamp(torch.as_tensor(action, dtype=torch.float32, device=self.device), -1, 1)
        roll, pitch, yaw = self.model.get_posture()
        vt = self.model.get_vt()
        tgt3 = torch.stack((pitch + action[:, 0] * 0.3, yaw + action[:, 1] * 0.3, vt + action[:, 2] * 30))  # :150-152
        masks = torch.ones((self.n, 1), device=self.device)
        for _ in range(INNER_STEPS):
            ego_obs = b.lowlevel_obs(tgt3)
            with torch.no_grad():
                ego_actions, _, self.ego_rnn_states = self.controller(ego_obs, self.ego_rnn_states, masks, deterministic=True)
            obs, reward, flags = b.step(ego_actions, inner=True)
            if render:
                self.render(count=count)
        f = flags.view(torch.bool)
        return obs, reward, f[0], f[1], f[2], self.info()
