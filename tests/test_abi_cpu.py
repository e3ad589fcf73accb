"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the public
header declares, its structs have the layout the ctypes binding assumes, and every compute entry
point FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'neuralplane_amd.h')


@pytest.fixture(scope='module')
def lib():
    from neuralplane_amd import _lib, build
    build.build_hip()
    return _lib.load()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(np_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol(lib):
    from neuralplane_amd import _lib
    names = _declared_functions()
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.np_abi_version() == _lib.ABI_VERSION == 13


def _flat_fields(struct, prefix=''):
    out = []
    for name, typ in struct._fields_:
        if isinstance(typ, type) and issubclass(typ, C.Structure):
            base = getattr(struct, name).offset
            out += [(prefix + name + '.' + n, base + o) for n, o in _flat_fields(typ)]
        else:
            out.append((prefix + name, getattr(struct, name).offset))
    return out


def test_struct_layout_matches_ctypes(tmp_path):
    """Compile a tiny C program against the public header and compare sizeof/offsetof with ctypes."""
    from neuralplane_amd import _lib
    pairs = [('np_f16_cfg', _lib.NpF16Cfg), ('np_f16_io', _lib.NpF16Io), ('np_pid_gains', _lib.NpPidGains),
             ('np_f16_combat_cfg', _lib.NpF16CombatCfg), ('np_f16_combat_io', _lib.NpF16CombatIo), ('np_planning_loop', _lib.NpPlanningLoop)]
    prog = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    exp = []
    for cname, st in pairs:
        prog.append(f'printf("%zu\\n", sizeof({cname}));')
        exp.append(C.sizeof(st))
        for f, off in _flat_fields(st):
            prog.append(f'printf("%zu\\n", offsetof({cname}, {f}));')
            exp.append(off)
    prog += ['return 0;}']
    c = tmp_path / 'layout.c'
    c.write_text('\n'.join(prog))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-std=c11', '-o', str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in out] == exp


def test_header_cites_reference_interfaces():
    src = open(HEADER).read()
    for cite in ('envs/env_base.py:83-97', 'envs/env_base.py:99-109', 'F16_model.py:51-67', 'hifi_F16_AeroData.py:40-129'):
        assert cite in src


def test_ctx_create_validates_blob_and_fails_loudly_without_gpu(lib):
    from neuralplane_amd import _lib
    from neuralplane_amd.core import ASSET_BLOB, cfg_from_config
    from neuralplane_amd.envs.utils.utils import parse_config
    cfg = cfg_from_config(parse_config('heading'), 'heading')
    ctx = C.c_void_p()
    # corrupt blob: rejected before any device is touched
    assert lib.np_f16_ctx_create(b'garbage' * 10, 70, C.byref(cfg), 0, C.byref(ctx)) != 0
    assert b'magic' in lib.np_last_error()
    blob = open(ASSET_BLOB, 'rb').read()
    bad = bytearray(blob)
    bad[16 + 128 * 3 + 24 + 4] = 9  # n_linear of net 3
    assert lib.np_f16_ctx_create(bytes(bad), len(bad), C.byref(cfg), 0, C.byref(ctx)) != 0
    assert b'class' in lib.np_last_error() or b'shape' in lib.np_last_error()
    import torch
    if not torch.cuda.is_available():
        rc = lib.np_f16_ctx_create(blob, len(blob), C.byref(cfg), 0, C.byref(ctx))
        assert rc != 0 and not ctx.value
        assert len(lib.np_last_error()) > 0
        with pytest.raises(RuntimeError):
            from neuralplane_amd.envs.control_env import ControlEnv
            ControlEnv(num_envs=4, config='heading', model='F16', random_seed=0, device='cuda:0')
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            ControlEnv(num_envs=4, config='heading', model='F16', random_seed=0, device='cpu')
        from neuralplane_amd.core import combat_cfg_from_config
        ccfg = combat_cfg_from_config(parse_config('selfplay'))
        assert lib.np_f16_combat_ctx_create(blob, len(blob), C.byref(ccfg), 0, C.byref(ctx)) != 0 and not ctx.value
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            from neuralplane_amd.envs.singlecombat_env import SingleCombatEnv
            SingleCombatEnv(num_envs=2, config='selfplay', random_seed=0, device='cpu')


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under neuralplane_amd/ or include/ may reference it."""
    bad = []
    for base in ('neuralplane_amd', 'include'):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith(('.py', '.h', '.hip', '.cpp', '.c')):
                    txt = open(os.path.join(dp, f), errors='ignore').read()
                    if re.search(r'f16_oracle|f16o_|from oracle|import oracle|oracle/', txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
    so = os.path.join(ROOT, 'neuralplane_amd', 'csrc', 'libneuralplane_hip.so')
    syms = subprocess.run(['nm', '-D', so], capture_output=True, text=True).stdout
    assert 'f16o_' not in syms
