"""Build the HIP extension in-tree: neuralplane_amd/csrc/libneuralplane_hip.so (gfx950 only).

    python -m neuralplane_amd.build          # or neuralplane_amd.build.build_hip()

hipcc cross-compiles for gfx950 without a GPU present.  -ffp-contract=off is part of the numerics
contract: the only fused operations are the explicit fmaf()/fma() calls.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(CSRC, 'libneuralplane_hip.so')
# one translation unit per task x solver for the env.step / env.reset kernels (np_env_tu.inc): the six compile side by side (the rk4 units
# are the long ones: ~2 min each), beside the host side + SingleCombat / controller kernels, the persistent PlanningEnv kernel and the dual family
SOURCES = ['np_env_t0s1.hip', 'np_env_t1s1.hip', 'np_env_t2s1.hip', 'np_env_t0s0.hip', 'np_env_t1s0.hip', 'np_env_t2s0.hip',
           'np_f16_kernels.hip', 'np_planning.hip', 'np_combat_lat.hip', 'np_actor_i8.hip', 'np_policy.hip']
HEADERS = ['np_f16_device.h', 'np_f16_kargs.h', 'np_f16_env_kernel.h', 'np_env_launch.h', 'np_env_tu.inc', 'np_planning.h', 'np_dispatch.h', 'np_f16_combat.h', 'np_actor.h', 'np_actor_i8.h', 'np_rollout.h', 'np_policy.h', 'np_actor_asm.inc', 'np_actor_mfma_asm.inc', 'np_actor_mfma16_asm.inc', 'np_math.h', 'np_nets.h', 'np_mlp_asm.inc', 'np_mlp_asm_dual.inc', os.path.join('..', '..', 'include', 'neuralplane_amd.h')]
# -disable-machine-licm: the SingleCombat kernel's inner loop (5 FDM steps) otherwise gets ~40 loop-invariant 64-bit constants of the
# fp64 sin / cos / pow sequences hoisted into VGPR pairs that stay live across the asm phases — 256 VGPRs plus 9 spilled dwords;
# re-materialised inside the loop it needs 200 and no scratch (the other kernels have no loops and compile identically)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-mllvm', '-disable-machine-licm']
OBJ_DIR = os.path.join(CSRC, 'build')


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP extension cannot be built (there is no CPU fallback)')


def _deps():
    return [os.path.join(CSRC, f) for f in HEADERS] + [os.path.abspath(__file__)]


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in _deps() + [os.path.join(CSRC, f) for f in SOURCES])


def build_hip(force=False, verbose=False):
    """Compile every HIP source for gfx950 (one object per translation unit, in parallel) and link libneuralplane_hip.so.
    Returns the path.  An object is rebuilt when its source, any header or this script is newer."""
    if not force and not is_stale():
        return SO
    extra = os.environ.get('NPF16_EXTRA_FLAGS', '').split()  # tuning experiments only (e.g. -DNPF16_BLOCK=64)
    os.makedirs(OBJ_DIR, exist_ok=True)
    stamp = os.path.join(OBJ_DIR, 'flags.txt')
    flags_now = ' '.join(FLAGS + extra)
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags_now
    newest_dep = max(os.path.getmtime(d) for d in _deps())
    jobs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + '.o')
        fresh = (not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= max(newest_dep, os.path.getmtime(path)))
        if fresh:
            continue
        cmd = [_hipcc()] + FLAGS + extra + ['-c', '-o', obj, path]
        if verbose:
            print(' '.join(cmd))
        jobs.append((src, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    errors = []
    for src, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            errors.append(f'{src}:\n{out}')
    if errors:
        raise RuntimeError('hipcc failed:\n' + '\n'.join(errors))
    with open(stamp, 'w') as f:
        f.write(flags_now)
    objs = [os.path.join(OBJ_DIR, os.path.splitext(src)[0] + '.o') for src in SOURCES]
    cmd = [_hipcc(), '--offload-arch=gfx950', '-fPIC', '-shared', '-o', SO] + objs
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc (link) failed:\n' + r.stdout)
    return SO


if __name__ == '__main__':
    import sys
    print(build_hip(force='--incremental' not in sys.argv, verbose=True))
