#!/usr/bin/env python3
"""tests/golden/ref_api_surface.json: the NAMES the reference's env-side modules define (classes, their methods, module-level
functions) — what a caller written against the reference can reach.  Names only, read with `ast` from /root/reference in the
build container; tests/test_host_logic_cpu.py checks the mirror under neuralplane_amd/envs against it.

    python tools/gen_api_surface.py
"""
import ast
import json
import os

REF = '/root/reference'
FILES = ['envs/env_base.py', 'envs/control_env.py', 'envs/planning_env.py', 'envs/singlecombat_env.py', 'envs/env_wrappers.py',
         'envs/models/model_base.py', 'envs/models/F16_model.py', 'envs/models/F16/F16_dynamics.py', 'envs/tasks/task_base.py',
         'envs/tasks/heading_task.py', 'envs/tasks/control_task.py', 'envs/tasks/tracking_task.py', 'envs/utils/utils.py',
         'envs/termination_conditions/termination_condition_base.py'] + [
    f'envs/termination_conditions/{m}.py' for m in ('overload', 'low_altitude', 'high_speed', 'low_speed', 'extreme_state', 'unreach_heading',
                                                   'unreach_posture', 'unreach_target', 'timeout')] + [
    f'envs/reward_functions/{m}.py' for m in ('reward_function_base', 'heading_reward', 'posture_reward', 'position_reward', 'event_driven_reward')]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'ref_api_surface.json')


def surface(path):
    tree = ast.parse(open(path).read())
    public = lambda name: not name.startswith('_') or name.startswith('__')      # noqa: E731  (private helpers are not surface)
    classes = {n.name: sorted(m.name for m in n.body if isinstance(m, ast.FunctionDef) and public(m.name))
               for n in tree.body if isinstance(n, ast.ClassDef)}
    functions = sorted(n.name for n in tree.body if isinstance(n, ast.FunctionDef) and public(n.name))
    return {'classes': classes, 'functions': functions}


if __name__ == '__main__':
    out = {f: surface(os.path.join(REF, f)) for f in FILES}
    json.dump(out, open(OUT, 'w'), indent=1, sort_keys=True)
    print('wrote', OUT, sum(len(v['functions']) + sum(len(m) for m in v['classes'].values()) for v in out.values()), 'names')
