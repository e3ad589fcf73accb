"""PositionReward — -0.1 x squared distance to the target point in km (position_reward.py:26-34); evaluated inside the step kernel, read back per aircraft."""
from .reward_function_base import _TaskTerm


class PositionReward(_TaskTerm):
    pass
