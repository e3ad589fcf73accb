"""Timeout — `step_count` reached `max_steps` (timeout.py): the step kernel reports it as the third flag, exceed_time_limit."""
import torch

from .termination_condition_base import BaseTerminationCondition


class Timeout(BaseTerminationCondition):
    def get_termination(self, task, env, info={}):  # noqa: B006
        t = env.exceed_time_limit
        zero = torch.zeros_like(t)
        return zero, zero, t.clone(), info
