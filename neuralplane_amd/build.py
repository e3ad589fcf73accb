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
SOURCES = ['np_f16_kernels.hip']
HEADERS = ['np_f16_device.h', 'np_f16_combat.h', 'np_actor.h', 'np_rollout.h', 'np_actor_asm.inc', 'np_actor_mfma_asm.inc', 'np_actor_mfma16_asm.inc', 'np_math.h', 'np_nets.h', 'np_mlp_asm.inc', 'np_mlp_asm_dual.inc', os.path.join('..', '..', 'include', 'neuralplane_amd.h')]
# -disable-machine-licm: the SingleCombat kernel's inner loop (5 FDM steps) otherwise gets ~40 loop-invariant 64-bit constants of the
# fp64 sin / cos / pow sequences hoisted into VGPR pairs that stay live across the asm phases — 256 VGPRs plus 9 spilled dwords;
# re-materialised inside the loop it needs 200 and no scratch (the other kernels have no loops and compile identically)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fPIC', '-shared', '-mllvm', '-disable-machine-licm']


def _hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the HIP extension cannot be built (there is no CPU fallback)')


def is_stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False):
    """Compile every HIP source for gfx950 into libneuralplane_hip.so.  Returns the path."""
    if not force and not is_stale():
        return SO
    extra = os.environ.get('NPF16_EXTRA_FLAGS', '').split()  # tuning experiments only (e.g. -DNPF16_BLOCK=64)
    cmd = [_hipcc()] + FLAGS + extra + ['-o', SO] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout)
    return SO


if __name__ == '__main__':
    print(build_hip(force=True, verbose=True))
