"""LowSpeed — Mach number below `min_velocity` (low_speed.py); evaluated inside the step kernel, read back per aircraft."""
from .termination_condition_base import BITS, BaseTerminationCondition


class LowSpeed(BaseTerminationCondition):
    bit = BITS['low_speed']
    kind = 'bad'
