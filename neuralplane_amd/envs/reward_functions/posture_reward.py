"""PostureReward — -(delta pitch / pi)^2 - (delta heading / pi)^2 - (delta vt / 340 m/s)^2 (posture_reward.py:26-35); evaluated inside the step kernel, read back per aircraft."""
from .reward_function_base import _TaskTerm


class PostureReward(_TaskTerm):
    pass
