"""UnreachHeading — heading task: target heading / altitude / vt not reached by `max_check_interval` -> bad_done; reached between `min_check_interval` and that -> done (unreach_heading.py:22-60); evaluated inside the step kernel, read back per aircraft."""
import torch

from .termination_condition_base import BITS, BaseTerminationCondition


class UnreachHeading(BaseTerminationCondition):
    def __init__(self, config, device=None):
        super().__init__(config)
        self.device = device

    def get_termination(self, task, env, info={}):  # noqa: B006
        reasons = env.termination_reasons()
        bad_done = (reasons >> BITS['unreach']) & 1 != 0
        done = (reasons >> BITS['reached']) & 1 != 0
        return bad_done, done, torch.zeros_like(done), info
