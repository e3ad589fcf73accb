"""FusedActor — PlanningEnv's frozen low-level controller as ONE HIP kernel launch per inner iteration.

The reference runs `PPOActor.forward(obs, rnn_states, masks, deterministic=True)` (algorithms/ppo/ppo_actor.py:38-64; the
configuration of envs/planning_env.py:18-29: feature LayerNorm, MLP 22-128-128 + ReLU + LayerNorm, GRU 128 + LayerNorm, act MLP
128-128, DiagGaussian mean head + tanh) 50 times per PlanningEnv.step: ~15 small torch kernels each, 151 K multiply-adds per
aircraft — nine times the FDM step it feeds.  `FusedActor` takes that actor's `state_dict()`, packs it once into a flat
kernel-order weight buffer and exposes the same call signature, so it drops into `PlanningEnv(controller=...)` (and into its
HIP-graph replay).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

OBS, HID, ACT = 22, 128, 4
# (name in the packed buffer, state_dict key, transpose?)  — kernel order; Linear weights are stored k-major (W^T)
_LAYOUT = [
    ('ln0_g', 'base.feature_norm.weight', False), ('ln0_b', 'base.feature_norm.bias', False),
    ('l1_b', 'base.mlp.fc.0.bias', False), ('l1_w', 'base.mlp.fc.0.weight', True),
    ('ln1_g', 'base.mlp.fc.2.weight', False), ('ln1_b', 'base.mlp.fc.2.bias', False),
    ('l2_b', 'base.mlp.fc.3.bias', False), ('l2_w', 'base.mlp.fc.3.weight', True),
    ('ln2_g', 'base.mlp.fc.5.weight', False), ('ln2_b', 'base.mlp.fc.5.bias', False),
    ('gi_b', 'rnn.gru.bias_ih_l0', False), ('gi_w', 'rnn.gru.weight_ih_l0', True),
    ('gh_b', 'rnn.gru.bias_hh_l0', False), ('gh_w', 'rnn.gru.weight_hh_l0', True),
    ('ln3_g', 'rnn.norm.weight', False), ('ln3_b', 'rnn.norm.bias', False),
    ('a1_b', 'act.mlp.fc.0.bias', False), ('a1_w', 'act.mlp.fc.0.weight', True),
    ('ln4_g', 'act.mlp.fc.2.weight', False), ('ln4_b', 'act.mlp.fc.2.bias', False),
    ('a2_b', 'act.mlp.fc.3.bias', False), ('a2_w', 'act.mlp.fc.3.weight', True),
    ('ln5_g', 'act.mlp.fc.5.weight', False), ('ln5_b', 'act.mlp.fc.5.bias', False),
    ('hd_b', 'act.action_out.mu_net.fc.0.bias', False), ('hd_w', 'act.action_out.mu_net.fc.0.weight', True),
]
_SHAPES = {'base.feature_norm.weight': (OBS,), 'base.mlp.fc.0.weight': (HID, OBS), 'base.mlp.fc.3.weight': (HID, HID),
           'rnn.gru.weight_ih_l0': (3 * HID, HID), 'rnn.gru.weight_hh_l0': (3 * HID, HID), 'act.mlp.fc.0.weight': (HID, HID),
           'act.mlp.fc.3.weight': (HID, HID), 'act.action_out.mu_net.fc.0.weight': (ACT, HID)}
NUM_FLOATS = 153392
NUM_FLOATS_I8 = 309312   # NP_ACTOR_I8_NUM_FLOATS: the same floats + per-output scales + limb fragments (np_actor_pack_i8)


SUPPORTED = 'hidden-size "128 128", act-hidden-size "128 128", recurrent-hidden-size 128 x 1 layer, feature LayerNorm, ReLU'


def reject_deeper_networks(state_dict, key_of=lambda k: k):
    """The reference builds its networks from --hidden-size / --act-hidden-size / --recurrent-hidden-size / --recurrent-hidden-layers
    (/root/reference/config.py:48-285): a third MLP layer or a second GRU layer shows up as extra state_dict keys, not as a shape."""
    for k in ('base.mlp.fc.6.weight', 'act.mlp.fc.6.weight', 'rnn.gru.weight_ih_l1'):
        if key_of(k) in state_dict:
            raise ValueError(f'{key_of(k)} present: a deeper network than the fused kernels are built for (supported: {SUPPORTED})')


def pack_ppo_actor(state_dict):
    """PPOActor.state_dict() (tensors or arrays) -> float32[153392] in kernel order.  Raises on any other architecture."""
    reject_deeper_networks(state_dict)
    parts = []
    for _, key, transpose in _LAYOUT:
        if key not in state_dict:
            raise ValueError(f'not a PlanningEnv low-level PPOActor state_dict: missing {key}')
        v = state_dict[key]
        v = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
        v = np.asarray(v, dtype=np.float32)
        if key in _SHAPES and tuple(v.shape) != _SHAPES[key]:
            raise ValueError(f'{key}: shape {tuple(v.shape)}, expected {_SHAPES[key]} (hidden 128 128, GRU 128 x 1, 22 obs, 4 actions)')
        parts.append(np.ascontiguousarray(v.T if transpose else v).reshape(-1))
    out = np.concatenate(parts)
    assert out.size == NUM_FLOATS, out.size
    return out


def pack_i8(packed_fp32):
    """float32[NUM_FLOATS] -> float32[NUM_FLOATS_I8]: the weight buffer of the block-fixed-point numerics (library call np_actor_pack_i8, host side)."""
    w = np.ascontiguousarray(packed_fp32, dtype=np.float32).reshape(-1)
    if w.size != NUM_FLOATS:
        raise ValueError(f'packed actor weights must hold {NUM_FLOATS} floats, got {w.size}')
    out = np.empty(NUM_FLOATS_I8, np.float32)
    _lib.check(_lib.load().np_actor_pack_i8(w.ctypes.data, out.ctypes.data))
    return out


class FusedActor:
    """`FusedActor(actor.state_dict(), device)(obs, rnn_states, masks, deterministic=True) -> (actions, None, rnn_states)`.
    numerics: 'fp32' (ordered fmaf chains, the CPU restatement f16_actor.inc) or 'i8' (block fixed point on the i8 matrix pipe, the CPU restatement f16_actor_i8.inc)."""

    def __init__(self, state_dict_or_packed, device='cuda:0', numerics='fp32'):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError(f"neuralplane_amd runs on MI355X (torch device 'cuda:N'), not on '{device}': there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        w = state_dict_or_packed
        if isinstance(w, dict):
            w = pack_ppo_actor(w)
        w = np.ascontiguousarray(w, dtype=np.float32).reshape(-1)
        if w.size != NUM_FLOATS:
            raise ValueError(f'packed actor weights must hold {NUM_FLOATS} floats, got {w.size}')
        if numerics not in ('fp32', 'i8'):
            raise ValueError("numerics: 'fp32' or 'i8'")
        self.numerics = numerics
        self.packed = w
        self.num_floats = NUM_FLOATS_I8 if numerics == 'i8' else NUM_FLOATS
        self.weights = torch.from_numpy(pack_i8(w) if numerics == 'i8' else w).to(self.device)

    @classmethod
    def from_checkpoint(cls, path, device='cuda:0', numerics='fp32'):
        """The reference's `actor_latest.pt` (a PPOActor state_dict saved by its runner, runner/F16sim_runner.py:223-229)."""
        sd = torch.load(path, map_location='cpu')
        if not isinstance(sd, dict):
            raise ValueError(f'{path}: expected a PPOActor state_dict')
        return cls({k: v for k, v in sd.items()}, device, numerics)

    def eval(self):
        return self

    def __call__(self, obs, rnn_states, masks, deterministic=True, out=None):
        """PPOActor.forward.  `out=(actions[n,4], rnn_states[n,1,128])`: write into caller-owned buffers (distinct from the
        inputs) instead of allocating — PlanningEnv ping-pongs two recurrent-state buffers through its 50 inner iterations."""
        if not deterministic:
            raise NotImplementedError('the frozen low-level controller acts deterministically (planning_env.py:158)')
        n = obs.shape[0]
        obs = obs.to(device=self.device, dtype=torch.float32).contiguous()
        h = rnn_states.to(device=self.device, dtype=torch.float32).reshape(n, HID).contiguous()
        if h.data_ptr() % 16:  # a view at an odd storage offset: the kernel reads the state 16 bytes at a time
            h = h.clone()
        m = masks.to(device=self.device, dtype=torch.float32).reshape(n).contiguous()
        if out is None:
            act = torch.empty((n, ACT), dtype=torch.float32, device=self.device)
            h_out = torch.empty((n, 1, HID), dtype=torch.float32, device=self.device)
        else:
            act, h_out = out
            ok = (act.is_contiguous() and h_out.is_contiguous() and act.dtype == torch.float32 and h_out.dtype == torch.float32 and
                  act.numel() == n * ACT and h_out.numel() == n * HID and act.device == self.device and h_out.device == self.device and
                  h_out.data_ptr() != h.data_ptr() and h_out.data_ptr() % 16 == 0)
            if not ok:
                raise ValueError('out = (actions[n,4], rnn_states[n,1,128]): contiguous float32 device tensors, the state buffer 16-byte aligned '
                                 'and different from the input state')
        stream = _lib.stream_ptr(self.device)
        _lib.check(self.lib.np_actor_forward(self.weights.data_ptr(), self.num_floats, n, obs.data_ptr(), h.data_ptr(), m.data_ptr(),
                                             act.data_ptr(), h_out.data_ptr(), self.device.index, stream))
        return act, None, h_out
