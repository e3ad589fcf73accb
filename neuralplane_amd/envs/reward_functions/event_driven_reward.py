"""EventDrivenReward — +200 for `done`, -200 for `bad_done` (event_driven_reward.py:16-29), from the flags of the last step."""
from .reward_function_base import BaseRewardFunction


class EventDrivenReward(BaseRewardFunction):
    def get_reward(self, task, env):
        return env.reward_terms()[1]
