"""The reference's reward-function objects (envs/reward_functions/*.py) on top of the fused step.

A task's reward is the sum of its reward functions (task_base.py:60-73): the task's own shaping term plus EventDrivenReward.  Both
are computed inside the step kernel; with np_f16_io.reward_task set the kernel also stores the task term alone, and these classes
return their term of the LAST step — same constructor, same `get_reward(task, env)`, no arithmetic of their own beyond the event
term's two constants.
"""


class BaseRewardFunction:
    def __init__(self, config):
        self.config = config

    def get_reward(self, task, env):
        raise NotImplementedError


class _TaskTerm(BaseRewardFunction):
    """The task's own reward function: what the step kernel computed for the state it reached."""

    def get_reward(self, task, env):
        return env.reward_terms()[0]
