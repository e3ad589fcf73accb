"""The reference's termination-condition objects (envs/termination_conditions/*.py) on top of the fused step.

In the reference every condition is a class whose `get_termination(task, env, info)` evaluates its test on the env's state and
returns `(bad_done, done, exceed_time_limit, info)`; the task ORs them (task_base.py:75-96).  Here all of them are evaluated
inside the step kernel; what it found per aircraft is kept as one byte of condition bits (np_f16_io.term_reasons), and these
classes read their bit of the LAST step back — same constructor, same method, same return shape, no arithmetic of their own.
"""
import torch

BITS = {'overload': 0, 'low_altitude': 1, 'high_speed': 2, 'low_speed': 3, 'extreme_state': 4, 'unreach': 5, 'reached': 6}


class BaseTerminationCondition:
    bit = None          # which bit of env.termination_reasons() this condition owns
    kind = 'bad'        # 'bad' -> bad_done, 'done' -> done

    def __init__(self, config):
        self.config = config

    def log(self, msg):
        pass

    def _mask(self, env):
        reasons = env.termination_reasons()
        return (reasons >> self.bit) & 1 != 0

    def get_termination(self, task, env, info={}):  # noqa: B006  (the reference's signature)
        """(bad_done, done, exceed_time_limit, info) of THIS condition at the state reached by the last env.step."""
        hit = self._mask(env)
        zero = torch.zeros_like(hit)
        return (hit, zero, zero, info) if self.kind == 'bad' else (zero, hit, zero, info)
