"""Overload — acceleration above `acceleration_limit` (overload.py:15-44); evaluated inside the step kernel, read back per aircraft."""
from .termination_condition_base import BITS, BaseTerminationCondition


class Overload(BaseTerminationCondition):
    bit = BITS['overload']
    kind = 'bad'
