"""SingleCombatEnv — 1v1 fly-combat env with the reference's surface (envs/singlecombat_env.py:25-274).

`SingleCombatEnv(num_envs=1, config='selfplay', random_seed=None, device='cuda:0')`: every env holds two
F-16s (rows 2k = ego, 2k+1 = enemy).  `step(action[n,4])` — throttle and roll/pitch/yaw demands per
aircraft — is ONE HIP kernel launch (np_f16_combat_step) that runs the pairwise auto-reset, the 5 inner
FDM steps behind the attitude PID stack, the terminations (Overload ... Crash, Timeout, Shutdown), the
15-float pairwise observation, the orientation x range reward and the blood update.

The reference file targets an older BaseEnv and cannot be constructed as shipped; DESIGN.md §10 lists
how its open ends were closed here (held policy action, controls written straight to `u`, terminations
after every inner FDM step, controller state kept across episode resets).
"""
import numpy as np
import torch

from ..core import F16CombatBatch, NUM_OBS_COMBAT
from .spaces import Box, Env
from .utils.utils import parse_config


class SingleCombatEnv(Env):
    def __init__(self, num_envs=1, config='selfplay', random_seed=None, device='cuda:0', env0=0, aero_1d_tables=None, airframe=None):
        super().__init__()
        self.config = parse_config(config)
        if airframe is not None:        # {np_f16_airframe field: value}: see BaseEnv
            self.config.airframe = dict(airframe)
        self.num_envs = num_envs
        self.num_agents = getattr(self.config, 'num_agents', 100)
        if self.num_agents != 2:
            raise NotImplementedError("Singlecombat number of agents must be 2!")
        self.n = self.num_agents * self.num_envs
        self.num_observation = getattr(self.config, 'num_observation', NUM_OBS_COMBAT)
        self.num_actions = getattr(self.config, 'num_actions', 4)
        if self.num_observation != NUM_OBS_COMBAT or self.num_actions != 4:
            raise NotImplementedError('SingleCombatEnv observes 15 floats and takes 4 actions per aircraft')
        self.dt = getattr(self.config, 'dt', 0.02)
        self.target_dist = getattr(self.config, 'target_dist', 3)
        self.observation_space = Box(low=-np.inf, high=np.inf, shape=(self.num_observation,))
        self.action_space = Box(low=-np.inf, high=np.inf, shape=(self.num_actions,))
        self.create_records = False
        seed = 0 if random_seed is None else int(random_seed)
        self._batch = F16CombatBatch(num_envs, self.config, device, seed=seed, env0=env0, aero_1d_tables=aero_1d_tables)
        self.device = self._batch.device

    # -- state as the reference exposes it ([n, k] views of the SoA buffers) -----------------------
    @property
    def s(self):
        return self._batch.s.t()

    @property
    def u(self):
        return self._batch.u.t()

    @property
    def blood(self):
        return self._batch.blood

    @property
    def step_count(self):
        return self._batch.step_count

    @property
    def is_done(self):
        return self._batch.flags[0].view(torch.bool)

    @property
    def bad_done(self):
        return self._batch.flags[1].view(torch.bool)

    @property
    def exceed_time_limit(self):
        return self._batch.flags[2].view(torch.bool)

    @property
    def controller_state(self):
        """[n, 11]: roll_dem, pitch_dem, then (error, integrator, last_out) of the roll / pitch / yaw rate PIDs."""
        return self._batch.pid.t()

    def info(self):
        return {}

    def get_number_of_agents(self):
        return self.n

    def seed(self, random_seed):
        self._batch.seed = int(random_seed) & 0xFFFFFFFFFFFFFFFF

    # -- pieces of the reference's step (singlecombat_env.py:60-181) -------------------------------------------------------------
    def obs(self):
        """Pairwise observation obs[n,15] of the current state: one kernel launch, nothing else changes."""
        return self._batch.observe()

    def reward(self):
        raise RuntimeError('SingleCombatEnv.reward is fused into step() (one HIP kernel): use the reward step() returns')

    def update_recent_s(self, s):
        """singlecombat_env.py:60-62 keeps the last two states for code that is commented out there (:141-142); kept as host
        bookkeeping only."""
        self.recent_s = [s, getattr(self, 'recent_s', [None, None])[0]]

    def termination_counts(self, reset=False):
        return self._batch.termination_counts(reset=reset)

    def state_dict(self):
        return self._batch.state_dict()

    def load_state_dict(self, sd):
        self._batch.load_state_dict(sd)

    # -- the hot path ------------------------------------------------------------------------------
    def reset(self, rand_u=None):
        """Re-initialise EVERY env (singlecombat_env.py:183-205) and return obs[n,15]."""
        self._batch.flags = torch.ones_like(self._batch.flags)
        return self._batch.reset(rand_u=rand_u)

    def reset_done_envs(self, rand_u=None):
        """Only reset envs in which an aircraft is flagged (singlecombat_env.py:207-238)."""
        return self._batch.reset(rand_u=rand_u)

    def step(self, action, rand_u=None):
        """(obs[n,15], reward[n], done, bad_done, exceed_time_limit, info) — singlecombat_env.py:240-274."""
        obs, reward, flags = self._batch.step(action, rand_u=rand_u)
        f = flags.view(torch.bool)
        return obs, reward, f[0], f[1], f[2], self.info()

    # -- the self-play runner's halves without the slicing (runner/selfplay_F16sim_runner.py:62-67, 96-100) ------------------------
    def reset_split(self, rand_u=None, out=None):
        """reset() returning (obs_ego[E,15], obs_opponent[E,15]) — `obs[:, :A//2]`, `obs[:, A//2:]` of the runner — as two
        contiguous arrays written by the kernel itself."""
        self._batch.flags = torch.ones_like(self._batch.flags)
        return self._batch.reset_split(rand_u=rand_u, out=out)

    def step_split(self, ego_action, opp_action, rand_u=None, out=None):
        """step() on (ego_action[E,4], opponent_action[E,4]) -> (obs_ego, obs_opponent, reward[n], done, bad_done,
        exceed_time_limit, info): the concatenate / split around env.step of the runner happens inside the kernel."""
        oe, oo, reward, flags = self._batch.step_split(ego_action, opp_action, rand_u=rand_u, out=out)
        f = flags.view(torch.bool)
        return oe, oo, reward, f[0], f[1], f[2], self.info()

    def render(self, count, filepath='./F16SimRecording.txt.acmi', env_index=0):
        """Append one TacView frame of engagement `env_index` (singlecombat_env.py:276-321): id 100 Red = ego, 101 Blue = enemy."""
        from .utils.acmi import AcmiRecorder
        if not self.create_records:
            self._acmi = AcmiRecorder(filepath)
            self.create_records = True
        rows = self._batch.s[:6, 2 * env_index:2 * env_index + 2].t().cpu().numpy()
        self._acmi.frame(count * self.dt, rows, colors=('Red', 'Blue'))
