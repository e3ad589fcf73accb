"""Scenario parsing and angle helpers of the env surface.

Mirrors the interface of the reference's envs/utils/utils.py: `parse_config` (:12-27) returns an
attribute bag whose keys are read with getattr(config, key, default); `wrap_PI` / `wrap_2PI`
(:144-154) are provided for callers that use them on tensors (PID, renders), as are the geodesy and pairwise
geometry helpers (geometry.py).  The hot path does not call these: the kernels carry their own bit-exact
versions (csrc/np_math.h, csrc/np_f16_combat.h).
"""
import os

import torch
import yaml

# WGS-84 conversions and the pairwise combat geometry / shaping functions of utils.py:35-250 (host-side helpers; the fused kernels
# evaluate the same geometry themselves)
from .geometry import (distance_fn, ecef_to_enu, ecef_to_geodetic, enu_to_ecef, enu_to_geodetic, geodetic_to_ecef,  # noqa: F401
                       geodetic_to_enu, get2d_AO_TA_R, get_AO_TA_R, orientation_fn, orientation_reward, range_reward)

CONFIG_DIR = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir, 'configs'))


class ScenarioConfig:
    """Attribute bag of one scenario YAML.  The env surface reads it the way the reference's callers do —
    `getattr(config, key, default)` and `config.init_state['init_T']` — so every top-level YAML key is an attribute."""

    def __init__(self, name, entries):
        self.__dict__.update(entries)
        self._scenario = name

    def __repr__(self):
        keys = sorted(k for k in self.__dict__ if not k.startswith('_'))
        return f'ScenarioConfig({self._scenario!r}: {", ".join(keys)})'


def get_root_dir():
    """Directory that holds `configs/` (the reference's contract: envs/)."""
    return os.path.dirname(CONFIG_DIR)


def parse_config(filename):
    """envs/configs/<filename>.yaml -> ScenarioConfig.  A scenario that does not exist is an AssertionError, as in the
    reference (envs/utils/utils.py:22-23), so callers that catch it keep working."""
    path = os.path.join(CONFIG_DIR, filename + '.yaml')
    if not os.path.isfile(path):
        raise AssertionError(f"no scenario '{filename}': {path} is missing (scenarios shipped: {sorted(available_configs())})")
    with open(path, encoding='utf-8') as fh:
        entries = yaml.safe_load(fh) or {}
    if not isinstance(entries, dict):
        raise AssertionError(f'{path}: a scenario file must be a YAML mapping')
    return ScenarioConfig(filename, entries)


def available_configs():
    return [os.path.splitext(f)[0] for f in os.listdir(CONFIG_DIR) if f.endswith('.yaml')]


def _t2n(tensor):
    """torch tensor -> numpy array on the host (the adapter GPUVecEnv uses)."""
    return tensor.detach().to('cpu').numpy()


def wrap_2PI(angle):
    res = angle % (2 * torch.pi)
    res = res + 2 * torch.pi * (res < 0)
    return res


def wrap_PI(angle):
    res = wrap_2PI(angle)
    res = res - 2 * torch.pi * (res > torch.pi)
    return res
