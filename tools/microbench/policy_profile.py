#!/usr/bin/env python3
"""A bare loop of FusedPolicy.get_actions for rocprofv3 (kernel stats / PMC passes): python tools/microbench/policy_profile.py n reps [NUMERICS=i8|fp32]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuralplane_amd.policy import FusedPolicy  # noqa: E402
from tools.collect_loop import TorchPolicy  # noqa: E402

n, reps = int(sys.argv[1]), int(sys.argv[2])
torch.manual_seed(0)
fp = FusedPolicy(TorchPolicy().eval().state_dicts(), 'cuda:0', numerics=os.environ.get('NUMERICS', 'i8'))
obs = torch.randn((n, 22), device='cuda:0')
ha, hc = torch.randn((n, 128), device='cuda:0') * 0.3, torch.randn((n, 128), device='cuda:0') * 0.3
m, eps = torch.ones((n, 1), device='cuda:0'), torch.randn((n, 4), device='cuda:0')
for _ in range(reps):
    fp.get_actions(obs, ha, hc, m, noise=eps)
torch.cuda.synchronize()
print('done', n, reps, fp.numerics)
