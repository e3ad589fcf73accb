"""HighSpeed — Mach number above `max_velocity` (high_speed.py); evaluated inside the step kernel, read back per aircraft."""
from .termination_condition_base import BITS, BaseTerminationCondition


class HighSpeed(BaseTerminationCondition):
    bit = BITS['high_speed']
    kind = 'bad'
