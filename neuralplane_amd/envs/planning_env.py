"""PlanningEnv — hierarchical tracking env with the reference's surface (envs/planning_env.py:32-177).

One high-level action (Δpitch, Δheading, Δvt) per `step`; inside, 50 low-level iterations of
{low-level observation -> frozen recurrent controller -> fused FDM step}.  Each iteration is ONE
kernel launch of this library (np_f16_step with inner_step=1, which also writes the low-level
observation of the state it reaches: np_f16_io.ll_obs; np_f16_lowlevel_obs runs once per macro-step,
for the first iteration) plus the controller's forward — 102 launches per step with the fused
controller (round 2: 151), the 100 of the loop enqueued by one library call (np_planning_inner_loop:
`_step_fused`; two or three row groups on their own streams for 8 192 < n <= 81 920);
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

from .. import _lib
from ..actor import FusedActor
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


class Args(_ActorArgs):  # the reference's name for the same object (planning_env.py:18: built for the CPU by default)
    def __init__(self, device=torch.device('cpu')):
        super().__init__(device)


class PlanningEnv(BaseEnv):
    def __init__(self, num_envs=1, config='tracking', model='F16', random_seed=None, device='cuda:0', controller=None,
                 controller_checkpoint=None, row0=0, aero_1d_tables=None, controller_numerics='i8', weights=None, airframe=None):
        super().__init__(num_envs, config, model, random_seed, device, row0=row0, aero_1d_tables=aero_1d_tables, weights=weights, airframe=airframe)
        self.low_level_action_space = Box(low=-np.inf, high=np.inf, shape=(4,))
        if isinstance(controller, str):
            if controller != 'fused':
                raise ValueError("controller: a callable with the PPOActor signature, None (the host repo's PPOActor) or 'fused'")
            # the reference's checkpoint (planning_env.py:16,43) run by the fused MFMA kernel instead of ~15 torch kernels per call
            ckpt = controller_checkpoint or self._default_checkpoint()
            if not os.path.exists(ckpt):
                raise RuntimeError(f'low-level controller checkpoint {ckpt} not found (it is not part of the reference snapshot)')
            # controller_numerics: 'i8' = block fixed point on the i8 matrix pipe (the default: 30 % less time per step, as close to the
            # reference's recordings as the fp32 chains), 'fp32' = ordered fmaf chains
            controller = FusedActor.from_checkpoint(ckpt, self.device, numerics=controller_numerics)
        self.controller = controller if controller is not None else self._load_reference_actor(controller_checkpoint)
        self.ego_rnn_states = torch.zeros((self.n, 1, 128), device=self.device)
        self._graph_enabled = False
        self._graph = None

    def load(self, random_seed, config, model):
        if model != 'F16':
            raise NotImplementedError
        if config != 'tracking':
            raise NotImplementedError
        batch = self._make_batch(config, random_seed)
        self.model = F16Model(self.config, self.n, self.device, random_seed, batch)
        self.task = TrackingTask(self.config, self.n, self.device, random_seed, batch)

    @staticmethod
    def _default_checkpoint():
        return os.path.join(os.getcwd(), '..', 'scripts', 'runs', '2024-05-26_02-14-24_Control_control_ppo_v1', 'episode_249', 'actor_latest.pt')

    def _load_reference_actor(self, checkpoint):
        try:
            from algorithms.ppo.ppo_actor import PPOActor  # the host repo's own actor (out of the accelerated path)
        except Exception as e:  # pragma: no cover
            raise RuntimeError('PlanningEnv needs a low-level controller: pass controller=..., or run inside the '
                               'NeuralPlane repo so that algorithms.ppo.ppo_actor.PPOActor is importable') from e
        ckpt = checkpoint or self._default_checkpoint()
        if not os.path.exists(ckpt):
            raise RuntimeError(f'low-level controller checkpoint {ckpt} not found (it is not part of the reference snapshot)')
        actor = PPOActor(_ActorArgs(self.device), self.observation_space, self.low_level_action_space, device=self.device)
        actor.eval()
        actor.load_state_dict(torch.load(ckpt, map_location=self.device))
        return actor

    def low_level_obs(self, target_pitch, target_heading, target_vt):
        """22-float observation of the low-level controller (planning_env.py:60-142)."""
        return self._batch.lowlevel_obs(torch.stack((target_pitch, target_heading, target_vt)))

    def _step_fused(self, action):
        """PlanningEnv.step with the fused controller: reset, the first low-level observation and ONE library call for the 50 iterations
        (np_planning_inner_loop: `loop_mode` 'persistent' / 'queue' = one launch of the persistent kernel, 'launches' = 2 x 50 launches as
        1-4 row groups on their own streams — two for 8 192 < n <= 16 384, three to 26 624, four to 36 864, two to 53 248, three to
        81 920; 'auto' = the library chooses).  The same arithmetic on the same inputs as the launch-by-launch path below:
        bit-identical (tests/test_gpu_actor.py)."""
        b, n, d = self._batch, self.n, self.device
        b.reset(want_obs=False)                                    # self.reset()           :145
        action = torch.as_tensor(action, dtype=torch.float32, device=d)
        if action.dim() != 2 or action.stride(1) != 1:
            action = action.reshape(n, -1).contiguous()
        p = getattr(self, '_loop_buf', None)
        if p is None:
            p = self._loop_buf = {'ll': [None, torch.empty((n, 22), dtype=torch.float32, device=d)],
                                  'tgt3': torch.empty((3, n), dtype=torch.float32, device=d),
                                  'rnn': [torch.empty((n, 128), dtype=torch.float32, device=d), torch.empty((n, 128), dtype=torch.float32, device=d)],
                                  'masks': torch.ones(n, dtype=torch.float32, device=d), 'act': torch.empty((n, 4), dtype=torch.float32, device=d),
                                  'flags': torch.empty((3, n), dtype=torch.uint8, device=d)}
        # clamp, (pitch, yaw, vt) + action * (0.3, 0.3, 30)  (:146-152) and the controller's first observation: one launch
        tgt3 = p['tgt3']
        p['ll'][0] = b.planning_targets_obs(action, tgt3)
        h = self.ego_rnn_states
        if h.data_ptr() != p['rnn'][0].data_ptr():   # somebody replaced the recurrent state (load_state_dict, the caller): take it over
            p['rnn'][0].copy_(torch.as_tensor(h, dtype=torch.float32, device=d).reshape(n, 128))
        flags_scratch = p['flags'] if p['flags'].data_ptr() != b.flags.data_ptr() else torch.empty((3, n), dtype=torch.uint8, device=d)
        args = (self.controller.weights, p['ll'], p['rnn'], p['masks'], p['act'], tgt3, flags_scratch, INNER_STEPS)
        try:
            obs, reward, flags = b.planning_inner_loop(*args, groups=self.loop_groups, mode=self.LOOP_MODES[self.loop_mode], waves=self.loop_waves,
                                                       block=self.loop_block, check=self.LOOP_CHECKS[self.loop_check])
        except _lib.PlanningStalled as e:
            if not e.restored:
                raise
            # a bounded wait of the guest / queue schedule expired (include/neuralplane_amd.h, "Bounded waits"): the library ended the kernel
            # and restored every buffer it updates in place, so the same macro-step runs again launch by launch — same results, bit for bit
            import warnings
            warnings.warn(f'{e}; re-running this macro-step launch by launch', RuntimeWarning, stacklevel=3)
            self.loop_fallbacks += 1
            obs, reward, flags = b.planning_inner_loop(*args, groups=self.loop_groups, mode=self.LOOP_MODES['launches'])
        self.ego_rnn_states = p['rnn'][INNER_STEPS & 1].view(n, 1, 128)
        f = flags.view(torch.bool)
        return obs, reward, f[0], f[1], f[2], self.info()

    loop_groups = 0          # np_planning_loop.groups (0 = the library chooses)
    LOOP_MODES = {'auto': 0, 'launches': 1, 'persistent': 2, 'queue': 3, 'guests': 4, 'dual': 5}
    LOOP_CHECKS = {'sync': 0, 'deferred': 1}
    loop_check = 'sync'      # np_planning_loop.check (guest / queue schedules): 'sync' = the call waits for the launch; a stalled schedule is re-run launch by launch
    loop_fallbacks = 0       # how many macro-steps were re-run that way
    loop_mode = 'auto'       # np_planning_loop.mode: 'launches' = 2 x 50 launches, 'persistent' / 'queue' = ONE launch (np_planning.hip)
    loop_waves = 0           # persistent kernel: waves per 32-row tile (0 = the library chooses, 4, 8)
    loop_block = 0           # queue schedule: iterations per (tile, block) work item (0 = the library chooses)
    use_inner_loop = True    # False: the launch-by-launch path (tests compare the two)

    def step(self, action, render=False, count=0):
        # an explicit enable_graph() wins over the default single-call path (it used to be silently ignored)
        if self._graph_enabled and not render:
            return self._step_graph(action)
        if self.use_inner_loop and isinstance(self.controller, FusedActor) and not render:
            return self._step_fused(action)
        b = self._batch
        b.reset(want_obs=False)                                    # self.reset()           :145
        action = torch.clamp(torch.as_tensor(action, dtype=torch.float32, device=self.device), -1, 1)
        roll, pitch, yaw = self.model.get_posture()
        vt = self.model.get_vt()
        tgt3 = torch.stack((pitch + action[:, 0] * 0.3, yaw + action[:, 1] * 0.3, vt + action[:, 2] * 30))  # :150-152
        masks = torch.ones((self.n, 1), device=self.device)
        tgt3 = tgt3.contiguous()
        ego_obs = b.lowlevel_obs(tgt3)        # the controller's first input; every later one is written by the inner step itself
        for k in range(INNER_STEPS):
            with torch.no_grad():
                ego_actions, _, self.ego_rnn_states = self.controller(ego_obs, self.ego_rnn_states, masks, deterministic=True)
            last = k == INNER_STEPS - 1
            nxt = None if last else torch.empty((self.n, 22), dtype=torch.float32, device=self.device)
            # one launch: the FDM step + the low-level observation of the state it reaches (np_f16_io.ll_obs); the task observation
            # is only wanted from the last iteration (planning_env.py:153-176 overwrites it every iteration)
            obs, reward, flags = b.step(ego_actions, inner=True, ll_tgt=None if last else tgt3, ll_obs=nxt, want_obs=last or render)
            ego_obs = nxt
            if render:
                self.render(count=count)
                count += 1                                             # :171-173: one frame per inner iteration
        f = flags.view(torch.bool)
        return obs, reward, f[0], f[1], f[2], self.info()

    # -- the whole macro-step as ONE HIP graph -------------------------------------------------
    def enable_graph(self, enable=True):
        """Replay PlanningEnv.step (1 reset + 50 x {low-level obs, controller forward, fused inner step} = ~100 kernel
        launches of this library plus the controller's own) from a single captured HIP graph: the step is launch-bound
        for small and medium batches.  Requirements: the controller is a pure torch module on this device (capturable),
        and nobody replaces `env.ego_rnn_states` / the state tensors between steps (in-place edits are fine).
        Results are bit-identical to the eager path (same kernels, RNG counter kept on the device).  Takes precedence over the
        fused controller's single-call path (`use_inner_loop`), which needs no graph: with a FusedActor leave it off."""
        self._graph_enabled = bool(enable)
        if not enable:
            self._graph = None

    def _macro_body(self, g):
        b = self._batch
        b.launch_static(g['fa'], g['fb'], 0)                                                     # self.reset()
        action = torch.clamp(g['action'], -1, 1)
        tgt3 = torch.stack((b.s[4] + action[:, 0] * 0.3, b.s[5] + action[:, 1] * 0.3, b.s[6] + action[:, 2] * 30))
        fin, fout = g['fb'], g['fa']
        fused = isinstance(self.controller, FusedActor)   # writes into caller-owned buffers: no allocation, no state copy
        rnn_a, rnn_b = g['rnn'], g.get('rnn2')
        tgt3 = tgt3.contiguous()
        b.lowlevel_obs_into(tgt3, g['ll_obs'])       # the controller's first input; the inner steps write the following ones
        for k in range(INNER_STEPS):
            last = k == INNER_STEPS - 1
            if fused:
                ego_actions, _, _ = self.controller(g['ll_obs'], rnn_a, g['masks'], deterministic=True, out=(g['ll_act'], rnn_b))
                rnn_a, rnn_b = rnn_b, rnn_a          # INNER_STEPS is even: the state ends up in g['rnn'] again
            else:
                ego_actions, _, rnn = self.controller(g['ll_obs'], g['rnn'], g['masks'], deterministic=True)
                g['rnn'].copy_(rnn)
                ego_actions = ego_actions.to(torch.float32).contiguous()
            # the first inner step re-evaluates the cached coefficients: the caller may have edited `s` between steps
            b.launch_static(fin, fout, 1 + k, action=ego_actions, obs=g['obs'] if last else None, reward=g['reward'], inner=True, cache_valid=(k > 0),
                            ll_tgt=None if last else tgt3, ll_obs=None if last else g['ll_obs'])
            fin, fout = fout, fin
        g['fa'].copy_(fin)                       # 1 + 50 flips end in the other buffer; the next replay starts from `fa`
        b.call_base.add_(1 + INNER_STEPS)

    def _capture(self):
        b, d, n = self._batch, self.device, self.n
        g = {'action': torch.zeros((n, 3), dtype=torch.float32, device=d),
             'fa': torch.empty((3, n), dtype=torch.uint8, device=d), 'fb': torch.empty((3, n), dtype=torch.uint8, device=d),
             'll_obs': torch.empty((n, 22), dtype=torch.float32, device=d), 'obs': torch.empty((n, 22), dtype=torch.float32, device=d),
             'reward': torch.empty(n, dtype=torch.float32, device=d), 'masks': torch.ones((n, 1), device=d),
             'rnn': self.ego_rnn_states.detach().clone()}
        if isinstance(self.controller, FusedActor):   # second recurrent-state buffer and the low-level actions, written in place
            g['rnn2'] = torch.empty_like(g['rnn'])
            g['ll_act'] = torch.empty((n, 4), dtype=torch.float32, device=d)
        assert INNER_STEPS % 2 == 0
        # warm-up on a side stream (library handles, autotuning) with the env state saved and restored around it
        saved = (b.state_dict(), b.coef_cache.clone(), g['rnn'].clone(), b.term_counters.clone())
        g['fa'].copy_(b.flags)
        b.call_base.fill_(b.call_idx)
        side = torch.cuda.Stream(device=d)
        side.wait_stream(torch.cuda.current_stream(d))
        with torch.cuda.stream(side), torch.no_grad():
            self._macro_body(g)
        torch.cuda.current_stream(d).wait_stream(side)
        b.load_state_dict(saved[0])
        b.coef_cache.copy_(saved[1])
        g['rnn'].copy_(saved[2])
        b.term_counters.copy_(saved[3])          # the warm-up's terminations were counted by device atomics: not part of the run
        graph = torch.cuda.CUDAGraph()
        g['fa'].copy_(b.flags)
        b.call_base.fill_(b.call_idx)
        with torch.cuda.graph(graph), torch.no_grad():
            self._macro_body(g)
        # capture does not execute: the device counter and the flags are still those of the current state
        # the optional per-aircraft outputs are baked into the graph's launches: keep them alive as long as the graph is, and remember
        # which set of pointers it was captured with (F16Batch._io_epoch)
        g['term_reasons'], g['reward_task'] = b.term_reasons, b.reward_task
        self._graph, self._gbuf = graph, g
        self._graph_call_idx = b.call_idx
        self._graph_io_epoch = b._io_epoch

    def _step_graph(self, action):
        b = self._batch
        if getattr(self, '_graph', None) is not None and self._graph_io_epoch != b._io_epoch:
            # track_termination_reasons / track_reward_terms changed an output pointer since the capture: the old graph would keep
            # writing to (or never write) the old buffers
            self._graph = None
        if getattr(self, '_graph', None) is None:
            self._capture()
        g = self._gbuf
        if b.call_idx != self._graph_call_idx or b.flags.data_ptr() != g['fa'].data_ptr():
            # somebody stepped / reset / restored the batch outside the graph: re-synchronise the device-side inputs
            g['fa'].copy_(b.flags)
            b.call_base.fill_(b.call_idx)
        if self.ego_rnn_states.data_ptr() != g['rnn'].data_ptr():
            g['rnn'].copy_(self.ego_rnn_states)
        g['action'].copy_(torch.as_tensor(action, dtype=torch.float32, device=self.device))
        self._graph.replay()
        b.call_idx += 1 + INNER_STEPS
        self._graph_call_idx = b.call_idx
        b.flags = g['fa']
        b._cache_valid = True
        b._s_version = b.s._version
        b._version += 1
        self.ego_rnn_states = g['rnn']
        f = g['fa'].clone().view(torch.bool)
        return g['obs'].clone(), g['reward'].clone(), f[0], f[1], f[2], self.info()

    # -- checkpoint: the batch state plus the low-level controller's recurrent state (without it a restored env diverges) -----
    def state_dict(self):
        sd = super().state_dict()
        sd['ego_rnn_states'] = self.ego_rnn_states.detach().clone()
        return sd

    def load_state_dict(self, sd):
        if 'ego_rnn_states' not in sd:
            raise KeyError("PlanningEnv checkpoint without 'ego_rnn_states': the controller's recurrent state is part of the env state")
        rnn = sd['ego_rnn_states'].to(self.device)
        if tuple(rnn.shape) != tuple(self.ego_rnn_states.shape):
            raise ValueError(f'ego_rnn_states: checkpoint {tuple(rnn.shape)} vs env {tuple(self.ego_rnn_states.shape)}')
        super().load_state_dict(sd)
        # a fresh tensor: the graph path notices that it is not its own buffer and copies it in before the next replay
        self.ego_rnn_states = rnn.clone()
