"""BaseEnv — the gym-style reset()/step() surface of the reference (envs/env_base.py:12-109).

Same constructor, attributes and return values; the body of reset()/step() is ONE HIP kernel
launch each (np_f16_reset / np_f16_step) on the current PyTorch-ROCm stream, so callers never
need an explicit synchronisation and nothing in the step forces a device->host sync (the
reference syncs >= 8 times per step through torch.any()/print and torch.sum() sizes).
"""
import torch

from ..core import F16Batch
from .spaces import Env
from .utils.utils import parse_config


class BaseEnv(Env):
    def __init__(self, num_envs=10, config='heading', model='F16', random_seed=None, device='cuda:0', row0=0,
                 aero_1d_tables=None, solver=None, weights=None, airframe=None):
        """airframe (not a reference argument): {np_f16_airframe field: value} for every constant that differs from the F-16's — mass, Jx,
        Jy, Jz, Jxz, S, B, cbar, xcg, xcgr, Heng, g, ail_ref, rud_ref, the atmosphere and command scales (include/neuralplane_amd.h) —,
        or the scenario YAML's `airframe:` mapping; together with `weights` (another NPF16MLP blob of the same topology) that is another
        aircraft on the same kernels.  Parity of a non-F-16 airframe is unpinned: the reference has none."""
        super().__init__()
        self.config = parse_config(config)
        if airframe is not None:
            self.config.airframe = dict(airframe)
        self.num_envs = num_envs
        self.num_agents = getattr(self.config, 'num_agents', 100)
        self.n = self.num_agents * self.num_envs
        self.device = torch.device(device)
        self.create_records = False
        self._row0 = row0
        self._aero_1d_tables = aero_1d_tables  # numerics option; None -> scenario key / env var / off
        self._weights = weights                # None -> the shipped F-16 blob; else the path of another NPF16MLP blob (same topology)
        self._solver = solver                  # None -> the scenario's `solver` key ('euler' | 'rk4', F16_model.py:16)
        self.load(random_seed, config, model)

    def load(self, random_seed, config, model):
        raise NotImplementedError

    def _make_batch(self, task, random_seed):
        # random_seed=None: the reference leaves torch's global generator unseeded; here the
        # counter-based RNG simply needs a key
        seed = 0 if random_seed is None else int(random_seed)
        self._batch = F16Batch(self.n, self.config, task, self.device, seed=seed, row0=self._row0,
                               aero_1d_tables=self._aero_1d_tables, solver=self._solver,
                               **({'blob_path': self._weights} if self._weights else {}))
        self.device = self._batch.device
        return self._batch

    # -- flags / counters as the reference exposes them ------------------------------------------
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
    def observation_space(self):
        return self.task.observation_space

    @property
    def action_space(self):
        return self.task.action_space

    @property
    def num_observation(self):
        return self.task.num_observation

    @property
    def num_actions(self):
        return self.task.num_actions

    def info(self):
        return {}

    def get_number_of_agents(self):
        return self.n

    def seed(self, random_seed):
        """env_base.py:36-40 seeds the process-wide generators (torch, numpy, random); the env's own draws are counter-based and
        keyed by the same seed."""
        self.model.seed(random_seed)
        self._batch.seed = int(random_seed) & 0xFFFFFFFFFFFFFFFF

    # -- the three pieces BaseEnv.step is made of in the reference (env_base.py:58-72) --------------------------------------------
    def obs(self):
        """Observation of the current state (task.get_obs): one kernel launch, nothing else changes."""
        return self._batch.observe()

    def reward(self):
        raise RuntimeError('BaseEnv.reward is fused into step() (one HIP kernel): use the reward step() returns')

    def done(self, info=None):
        raise RuntimeError('BaseEnv.done is fused into step() (one HIP kernel): use the masks step() returns '
                           '(also kept as env.is_done / env.bad_done / env.exceed_time_limit)')

    def termination_counts(self, reset=False):
        """Per-condition termination statistics accumulated on the device (the reference prints them every step)."""
        return self._batch.termination_counts(reset=reset)

    def termination_reasons(self):
        """uint8[n]: which termination conditions fired for which aircraft at the state reached by the LAST step (bit k =
        F16Batch.TERM_NAMES[k]: overload, low_altitude, high_speed, low_speed, extreme_state, unreach, reached).  The first call
        switches the tracking on: the kernel then stores one more byte per aircraft and step; it reports from the next step on.
        PlanningEnv: the bits accumulate over the 50 inner iterations of one step (as its done / bad_done flags do), so a row that
        tripped Overload at inner iteration 3 still shows the bit after the step; reset() clears them."""
        r = self._batch.term_reasons
        return r if r is not None else self._batch.track_termination_reasons(True)

    def reward_terms(self):
        """(task term, event term) of the LAST step's reward, float32[n] each: the task's reward function (HeadingReward /
        PostureReward / PositionReward) and EventDrivenReward's -200 * bad_done + 200 * done; their fp32 sum is the reward step()
        returned.  The first call switches the tracking on (one more float stored per aircraft and step, from the next step on).
        PlanningEnv: both terms are those of the LAST of the 50 inner iterations — the one whose reward step() returns — with the
        event term on the flags accumulated over all 50 (planning_env.py:153-176)."""
        r = self._batch.reward_task
        if r is None:
            r = self._batch.track_reward_terms(True)
        f = self._batch.flags
        event = -200.0 * f[1].to(torch.float32) + 200.0 * f[0].to(torch.float32)
        return r, event

    def state_dict(self):
        """Env-state checkpoint (tensors on the env device); see F16Batch.state_dict."""
        return self._batch.state_dict()

    def load_state_dict(self, sd):
        self._batch.load_state_dict(sd)

    # -- the hot path ------------------------------------------------------------------------------
    def reset(self, rand_u=None, noise=None):
        """Re-initialise flagged rows, clear the flags, return obs[n,22] (env_base.py:83-97)."""
        return self._batch.reset(rand_u=rand_u, noise=noise)

    def step(self, action, render=False, count=0, rand_u=None, noise=None):
        """(obs, reward, done, bad_done, exceed_time_limit, info) — env_base.py:99-109.

        `rand_u` / `noise` are parity hooks (inject the reference's random draws); normally None.
        """
        obs, reward, flags = self._batch.step(action, rand_u=rand_u, noise=noise)
        if render:
            self.render(count=count)
        done, bad_done, exceed_time_limit = flags.view(torch.bool).unbind(0)
        return obs, reward, done, bad_done, exceed_time_limit, self.info()

    def render(self, count, filename='./tracks/F16SimRecording-', max_aircraft=64):
        """Append one TacView frame (env_base.py:111-151).  A new file `<filename><count>.txt.acmi` is started at
        count == 0 and after any aircraft terminated, as in the reference; ids 100+i, Name=F16, Color=Red.  The reference
        loop is only meaningful for n == 1; here the first `max_aircraft` rows are drawn."""
        from .utils.acmi import AcmiRecorder
        if count == 0:
            self.create_records = False
        if not self.create_records:
            self._acmi = AcmiRecorder(filename + str(count) + '.txt.acmi')
            self.create_records = True
        k = min(self.n, max_aircraft)
        rows = self._batch.s[:6, :k].t().cpu().numpy()            # one small D2H copy per rendered frame
        self._acmi.frame(float(self._batch.step_count[0].item()) * self.model.dt, rows)
        if bool(self._batch.flags.any().item()):
            self.create_records = False
