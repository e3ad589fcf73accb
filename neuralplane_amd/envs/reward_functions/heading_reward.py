"""HeadingReward — -(delta altitude [km])^2 - (delta heading / pi)^2 - (delta vt / 340 m/s)^2 (heading_reward.py:26-36); evaluated inside the step kernel, read back per aircraft."""
from .reward_function_base import _TaskTerm


class HeadingReward(_TaskTerm):
    pass
