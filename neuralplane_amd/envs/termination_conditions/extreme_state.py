"""ExtremeState — angle of attack / sideslip outside their limits (extreme_state.py); evaluated inside the step kernel, read back per aircraft."""
from .termination_condition_base import BITS, BaseTerminationCondition


class ExtremeState(BaseTerminationCondition):
    bit = BITS['extreme_state']
    kind = 'bad'
