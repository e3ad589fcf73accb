"""Scenario parsing and angle helpers of the env surface.

Mirrors the interface of the reference's envs/utils/utils.py: `parse_config` (:12-27) returns an
attribute bag whose keys are read with getattr(config, key, default); `wrap_PI` / `wrap_2PI`
(:144-154) are provided for callers that use them on tensors (PID, renders).  The hot path does
not call these: the kernels carry their own bit-exact wrap (csrc/np_math.h).
"""
import os

import torch
import yaml


def get_root_dir():
    return os.path.join(os.path.split(os.path.realpath(__file__))[0], '..')


def parse_config(filename):
    """Parse envs/configs/<filename>.yaml into an attribute bag (same contract as the reference)."""
    filepath = os.path.join(get_root_dir(), 'configs', f'{filename}.yaml')
    assert os.path.exists(filepath), \
        f'config path {filepath} does not exist. Please pass in a string that represents the file path to the config yaml.'
    with open(filepath, 'r', encoding='utf-8') as f:
        config_data = yaml.load(f, Loader=yaml.FullLoader)
    return type('EnvConfig', (object,), config_data)


def _t2n(x):
    return x.detach().cpu().numpy()


def wrap_2PI(angle):
    res = angle % (2 * torch.pi)
    res = res + 2 * torch.pi * (res < 0)
    return res


def wrap_PI(angle):
    res = wrap_2PI(angle)
    res = res - 2 * torch.pi * (res > torch.pi)
    return res
