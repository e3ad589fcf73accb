"""FusedPolicy — the rollout policy's inference calls as ONE HIP kernel launch each ("policy-side fusion at the boundary").

The reference's collect step (runner/F16sim_runner.py:123-129) is `PPOPolicy.get_actions(obs, rnn_states_actor, rnn_states_critic, masks)`
(algorithms/ppo/ppo_policy.py:26-32): PPOActor.forward with sampled actions + log-probabilities and PPOCritic.forward, ~110 small torch
kernels — 0.55-0.60 ms per step, nine tenths of a device-resident collect step (profiles/r05_collect_loop.json).  The training scripts
build both networks in the frozen controller's shapes (hidden "128 128", act-hidden "128 128", GRU 128 x 1, feature LayerNorm, ReLU;
22 observations — 15 for the 1v1 combat env's policies, runner/selfplay_F16sim_runner.py —; 4 actions for heading / control, 3 for tracking), so they run through the controller's matrix-core tile bodies
(csrc/np_policy.hip: fp32 chains; csrc/np_actor_i8.hip policy_act_i8_kernel: block fixed point on the i8 pipe, the default):
`FusedPolicy(policy)` packs `policy.actor` / `policy.critic` once and exposes the reference's three inference calls with its signatures

    get_actions(obs, rnn_states_actor, rnn_states_critic, masks) -> values, actions, action_log_probs, rnn_states_actor, rnn_states_critic
    get_values(obs, rnn_states_critic, masks)                    -> values
    act(obs, rnn_states_actor, masks, deterministic=False)       -> actions, rnn_states_actor

on device tensors.  Training (evaluate_actions, the optimiser) stays the host repository's torch code; after every update call
`refresh()` (or construct with `auto_refresh=True`: the parameters' version counters are checked at every call) to re-pack the weights.
The standard normal draws of a sampled step come from torch's generator on this device (`torch.randn`), as the reference's do.
There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .actor import _LAYOUT, _SHAPES, HID, NUM_FLOATS, NUM_FLOATS_I8, OBS, SUPPORTED, pack_i8, reject_deeper_networks

MAX_ACT = 4
_CRITIC_KEYS = {'act.mlp.fc.0': 'mlp.fc.0', 'act.mlp.fc.2': 'mlp.fc.2', 'act.mlp.fc.3': 'mlp.fc.3', 'act.mlp.fc.5': 'mlp.fc.5'}
_HEAD_W, _HEAD_B = 'act.action_out.mu_net.fc.0.weight', 'act.action_out.mu_net.fc.0.bias'


def _np(v):
    v = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
    return np.asarray(v, dtype=np.float32)


OBS_DIMS = (22, 15)   # envs/configs/*.yaml num_observation: control / heading / tracking 22, the 1v1 combat env (selfplay.yaml) 15


def _pack(state_dict, key_of, head_w, head_b, what):
    """-> (float32[NUM_FLOATS], obs_dim).  A network on fewer than 22 observations keeps the layout: its feature LayerNorm terms and the first
    layer's columns are zero-padded to 22 (the kernels read the first obs_dim of them)."""
    parts = []
    k0 = key_of('base.feature_norm.weight')
    if k0 not in state_dict:
        raise ValueError(f'not a {what} state_dict of the supported architecture: missing {k0}')
    reject_deeper_networks(state_dict, key_of)
    obs_dim = int(_np(state_dict[k0]).shape[0])
    if obs_dim not in OBS_DIMS:
        raise ValueError(f'{k0}: {obs_dim} observations; supported: {OBS_DIMS}')
    for _, key, transpose in _LAYOUT:
        if key == _HEAD_W:
            v = head_w
        elif key == _HEAD_B:
            v = head_b
        else:
            k = key_of(key)
            if k not in state_dict:
                raise ValueError(f'not a {what} state_dict of the supported architecture: missing {k}')
            v = _np(state_dict[k])
            want = {'base.feature_norm.weight': (obs_dim,), 'base.feature_norm.bias': (obs_dim,), 'base.mlp.fc.0.weight': (HID, obs_dim)}.get(key, _SHAPES.get(key))
            if want is not None and tuple(v.shape) != want:
                raise ValueError(f'{k}: shape {tuple(v.shape)}, expected {want} (supported: {SUPPORTED}; {obs_dim} observations)')
            if obs_dim != OBS and key in ('base.feature_norm.weight', 'base.feature_norm.bias', 'base.mlp.fc.0.weight'):
                pad = np.zeros(v.shape[:-1] + (OBS,), np.float32)
                pad[..., :obs_dim] = v
                v = pad
        parts.append(np.ascontiguousarray(v.T if transpose else v).reshape(-1))
    out = np.concatenate(parts)
    assert out.size == NUM_FLOATS, out.size
    return out, obs_dim


def pack_policy_actor(state_dict):
    """PPOActor.state_dict() with a DiagGaussian head of 1..4 actions -> (float32[NUM_FLOATS] in np_actor_forward's layout, the head
    zero-padded to four columns; act_dim; log_std float32[act_dim])."""
    for k in (_HEAD_W, _HEAD_B, 'act.action_out.log_std'):
        if k not in state_dict:
            raise ValueError(f'not a PPOActor state_dict with a DiagGaussian head: missing {k}')
    w, b, log_std = _np(state_dict[_HEAD_W]), _np(state_dict[_HEAD_B]), _np(state_dict['act.action_out.log_std']).reshape(-1)
    A = w.shape[0]
    if not (1 <= A <= MAX_ACT) or w.shape != (A, HID) or b.shape != (A,) or log_std.shape != (A,):
        raise ValueError(f'mu_net: {tuple(w.shape)} / {tuple(b.shape)} / log_std {tuple(log_std.shape)}; supported: 1..4 actions on 128 features ({SUPPORTED})')
    wp, bp = np.zeros((MAX_ACT, HID), np.float32), np.zeros(MAX_ACT, np.float32)
    wp[:A], bp[:A] = w, b
    w, _ = _pack(state_dict, lambda k: k, wp, bp, 'PPOActor')
    return w, A, log_std.copy()


def pack_policy_critic(state_dict):
    """PPOCritic.state_dict() (ppo_critic.py:10-36: base, rnn, mlp, value_out) -> float32[NUM_FLOATS]: the actor layout with `mlp` in the
    place of `act.mlp` and value_out in column 0 of the head block."""
    for k in ('value_out.weight', 'value_out.bias'):
        if k not in state_dict:
            raise ValueError(f'not a PPOCritic state_dict: missing {k}')
    w, b = _np(state_dict['value_out.weight']), _np(state_dict['value_out.bias'])
    if w.shape != (1, HID) or b.shape != (1,):
        raise ValueError(f'value_out: {tuple(w.shape)} / {tuple(b.shape)}, expected (1, 128) / (1,) (supported: {SUPPORTED})')
    wp, bp = np.zeros((MAX_ACT, HID), np.float32), np.zeros(MAX_ACT, np.float32)
    wp[0], bp[0] = w[0], b[0]

    def key_of(key):
        for a, c in _CRITIC_KEYS.items():
            if key.startswith(a + '.'):
                return c + key[len(a):]
        return key
    return _pack(state_dict, key_of, wp, bp, 'PPOCritic')[0]


def obs_dim_of(state_dict):
    """Observations per row of a PPOActor / PPOCritic state_dict (22 or 15)."""
    return int(_np(state_dict['base.feature_norm.weight']).shape[0])


class NpPolicyStep(C.Structure):   # include/neuralplane_amd.h: np_policy_step
    _fields_ = [('n', C.c_int64), ('act_dim', C.c_int32), ('flags', C.c_int32), ('actor_weights', C.c_void_p), ('critic_weights', C.c_void_p),
                ('std', C.c_float * 4), ('log_std', C.c_float * 4), ('obs', C.c_void_p), ('masks', C.c_void_p), ('noise', C.c_void_p),
                ('rnn_states_actor_in', C.c_void_p), ('rnn_states_critic_in', C.c_void_p), ('values', C.c_void_p), ('actions', C.c_void_p),
                ('action_log_probs', C.c_void_p), ('rnn_states_actor_out', C.c_void_p), ('rnn_states_critic_out', C.c_void_p), ('weights_floats', C.c_int64),
                ('obs_dim', C.c_int32), ('reserved_', C.c_int32), ('prev_flags', C.c_void_p), ('masks_out', C.c_void_p), ('bad_masks_out', C.c_void_p)]


ACTOR, CRITIC, DETERMINISTIC = 1, 2, 4


class FusedPolicy:
    """`FusedPolicy(policy)` for an object with `.actor` / `.critic` torch modules of PPOActor's / PPOCritic's structure (the reference's
    PPOPolicy), or `FusedPolicy((actor_state_dict, critic_state_dict))`.  See the module docstring."""

    def __init__(self, policy, device='cuda:0', auto_refresh=False, numerics='i8'):
        """numerics: 'i8' (default, as PlanningEnv's controller) = block fixed point on the i8 matrix pipe — against the reference's recording actions
        3.4e-6, values 2.6e-5, log-probabilities 1.9e-6; 22 / 49 / 360 us per get_actions at 3 000 / 10 000 / 100 000 rows; 'fp32' = ordered fmaf chains on the
        fp32 matrix pipe (actions 1.1e-6, values 1.4e-5; 34 / 83 / 642 us)."""
        if numerics not in ('fp32', 'i8'):
            raise ValueError("numerics: 'fp32' or 'i8'")
        self.numerics = numerics
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError(f"neuralplane_amd runs on MI355X (torch device 'cuda:N'), not on '{device}': there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self._source = policy
        self.auto_refresh = bool(auto_refresh)
        self.num_floats = NUM_FLOATS_I8 if numerics == 'i8' else NUM_FLOATS
        self.weights = torch.empty((2, self.num_floats), dtype=torch.float32, device=self.device)   # [0] actor, [1] critic; rows are 16-byte aligned
        assert self.num_floats % 4 == 0 and self.weights.data_ptr() % 16 == 0
        self._versions = None
        self.refreshes = 0
        self.refresh()

    def _state_dicts(self):
        p = self._source
        if isinstance(p, (tuple, list)):
            return p[0], p[1]
        return p.actor.state_dict(), p.critic.state_dict()

    def _watch(self):
        p = self._source
        if isinstance(p, (tuple, list)):
            return None
        return tuple(int(q._version) for net in (p.actor, p.critic) for q in net.parameters())

    def refresh(self):
        """Re-pack both networks from the source policy (after an optimiser step or load_state_dict)."""
        sa, sc = self._state_dicts()
        wa, self.act_dim, log_std = pack_policy_actor(sa)
        wc = pack_policy_critic(sc)
        self.obs_dim = obs_dim_of(sa)
        if obs_dim_of(sc) != self.obs_dim:
            raise ValueError(f'actor on {self.obs_dim} observations, critic on {obs_dim_of(sc)}')
        if self.numerics == 'i8':   # per-output scales + limb fragments behind the same floats (np_actor_pack_i8, host side)
            wa, wc = pack_i8(wa), pack_i8(wc)
        self.weights.copy_(torch.from_numpy(np.stack((wa, wc))))
        ls = torch.from_numpy(log_std)
        v = sa['act.action_out.log_std']
        # std as the reference computes it: exp() by torch on the parameter's own device (distributions.py:96)
        std = v.detach().exp().reshape(-1).to(torch.float32).cpu() if hasattr(v, 'detach') else ls.exp()
        self.log_std, self.std = [float(x) for x in ls], [float(x) for x in std]
        q = self._q = NpPolicyStep()
        q.act_dim, q.weights_floats, q.obs_dim = self.act_dim, self.num_floats, self.obs_dim
        q.actor_weights, q.critic_weights = self.weights[0].data_ptr(), self.weights[1].data_ptr()
        for j in range(self.act_dim):
            q.std[j], q.log_std[j] = self.std[j], self.log_std[j]
        self._versions = self._watch()
        self.refreshes += 1

    def _maybe_refresh(self):
        if self.auto_refresh and self._versions is not None and self._watch() != self._versions:
            self.refresh()

    def _rows(self, x, n, width):
        if type(x) is torch.Tensor and x.dtype is torch.float32 and x.device == self.device and x.is_contiguous() and x.numel() == n * width and x.data_ptr() % 16 == 0:
            return x   # the common case (a device-resident loop): only the address is used
        x = torch.as_tensor(x, device=self.device).to(dtype=torch.float32).reshape(n, width)
        x = x if x.is_contiguous() else x.contiguous()
        return x if x.data_ptr() % 16 == 0 else x.clone()

    def _launch(self, flags, obs, ha, hc, masks, noise):
        self._maybe_refresh()
        d = self.device
        if type(obs) is not torch.Tensor:
            obs = torch.as_tensor(obs, device=d)
        n = obs.shape[0]
        obs = self._rows(obs, n, self.obs_dim)
        m = self._rows(masks, n, 1)
        q = self._q   # weights, act_dim, std / log_std: filled by refresh(); the library reads the struct during the call only
        q.n, q.flags = n, flags
        q.obs, q.masks, q.prev_flags = obs.data_ptr(), m.data_ptr(), None   # (prev_flags: the collector's mode, collect.py)
        out = {}
        # every staged input stays referenced until the launch is enqueued: a copy made by _rows (numpy / non-contiguous input) that lost
        # its last reference before then would go back to the caching allocator, and the torch.empty of an OUTPUT of the same launch could
        # be handed its block (np_policy_act also rejects any input range that overlaps an output range)
        keep = [obs, m]
        if flags & ACTOR:
            h_a = self._rows(ha, n, HID)
            keep.append(h_a)
            if not flags & DETERMINISTIC:
                if noise is None:
                    noise = torch.randn((n, self.act_dim), dtype=torch.float32, device=d)
                noise = self._rows(noise, n, self.act_dim)
                keep.append(noise)
                q.noise = noise.data_ptr()
        if flags & CRITIC:
            h_c = self._rows(hc, n, HID)
            keep.append(h_c)
        if flags & ACTOR:
            out['actions'] = torch.empty((n, self.act_dim), dtype=torch.float32, device=d)
            out['logp'] = torch.empty((n, 1), dtype=torch.float32, device=d)
            out['ha'] = torch.empty((n, 1, HID), dtype=torch.float32, device=d)
            q.rnn_states_actor_in, q.rnn_states_actor_out = h_a.data_ptr(), out['ha'].data_ptr()
            q.actions, q.action_log_probs = out['actions'].data_ptr(), out['logp'].data_ptr()
        if flags & CRITIC:
            out['values'] = torch.empty((n, 1), dtype=torch.float32, device=d)
            out['hc'] = torch.empty((n, 1, HID), dtype=torch.float32, device=d)
            q.rnn_states_critic_in, q.rnn_states_critic_out, q.values = h_c.data_ptr(), out['hc'].data_ptr(), out['values'].data_ptr()
        _lib.check(self.lib.np_policy_act(C.byref(q), self.device.index, _lib.stream_ptr(d)))
        del keep   # enqueued: later allocations are ordered behind the kernel on this stream
        return out

    # ---- the reference's three inference calls (ppo_policy.py:26-57) ----
    def get_actions(self, obs, rnn_states_actor, rnn_states_critic, masks, noise=None):
        """-> values [n,1], actions [n,A], action_log_probs [n,1], rnn_states_actor [n,1,128], rnn_states_critic [n,1,128].  `noise` [n,A]: the
        standard normal draws to use instead of torch.randn (tests; replaying a recorded step)."""
        o = self._launch(ACTOR | CRITIC, obs, rnn_states_actor, rnn_states_critic, masks, noise)
        return o['values'], o['actions'], o['logp'], o['ha'], o['hc']

    def get_values(self, obs, rnn_states_critic, masks):
        return self._launch(CRITIC, obs, None, rnn_states_critic, masks, None)['values']

    def act(self, obs, rnn_states_actor, masks, deterministic=False, noise=None):
        o = self._launch(ACTOR | (DETERMINISTIC if deterministic else 0), obs, rnn_states_actor, None, masks, noise)
        return o['actions'], o['ha']

    def prep_rollout(self):
        p = self._source
        if hasattr(p, 'prep_rollout'):
            p.prep_rollout()

    def prep_training(self):
        p = self._source
        if hasattr(p, 'prep_training'):
            p.prep_training()
