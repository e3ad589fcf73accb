"""LowAltitude — altitude below `altitude_limit` (low_altitude.py); evaluated inside the step kernel, read back per aircraft."""
from .termination_condition_base import BITS, BaseTerminationCondition


class LowAltitude(BaseTerminationCondition):
    bit = BITS['low_altitude']
    kind = 'bad'
