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
    assert lib.np_abi_version() == _lib.ABI_VERSION == 16


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
             ('np_f16_combat_cfg', _lib.NpF16CombatCfg), ('np_f16_combat_io', _lib.NpF16CombatIo), ('np_planning_loop', _lib.NpPlanningLoop),
             ('np_dispatch_info', _lib.NpDispatchInfo), ('np_rollout_step', _lib.NpRolloutStep), ('np_policy_step', __import__('neuralplane_amd.policy', fromlist=['x']).NpPolicyStep)]
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


def test_dispatch_plan_scales_with_the_cu_count(lib):
    """np_dispatch_plan (csrc/np_dispatch.h): the variant / tiling / row-group selection is a pure function of (n, CU count).  On 256 CUs it
    reproduces the measured thresholds of the full MI355X exactly; on a partitioned device (CPX: 32 CUs, DPX: 128) the same decision is
    taken at the same number of rows PER CU (the literals used to assume 256 CUs)."""
    from neuralplane_amd import _lib
    P = _lib.dispatch_plan
    # ---- 256 CUs: the shipped choices, boundary by boundary (np_f16_step, Euler, MLP numerics)
    def fam(d):
        return ('pair3' if d['pair3'] else 'pair' if d['pair'] else 'lat8' if d['latency8'] else 'lat2' if d['latency2'] else
                'lat4w' if d['latency4w'] else 'lat4' if d['latency'] else 'thr')
    expect = [(1, 'lat8'), (16384, 'lat8'), (16385, 'lat4'), (49152, 'lat4'), (49153, 'lat4w'), (65536, 'lat4w'), (65537, 'lat2'), (98304, 'lat2'),
              (98305, 'pair'), (131072, 'pair'), (131073, 'pair3'), (1_000_000, 'pair3')]
    for n, f in expect:
        assert fam(P(n, 256)) == f, (n, f, P(n, 256))
    d = P(1_000_000, 256)
    assert d['grid'] == (1_000_000 + 127) // 128 and d['block'] == 128
    assert P(10_000, 256)['grid'] == 157 and P(10_000, 256)['block'] == 512 and P(30_000, 256)['block'] == 256 and P(80_000, 256)['block'] == 128
    # rk4 and the reset have no latency family; the table numerics no eight-wave / four-waves-per-SIMD builds
    assert fam(P(1000, 256, solver=1)) == 'thr' and fam(P(1000, 256, step=False)) == 'thr'
    assert fam(P(1000, 256, tables=True)) == 'lat4' and fam(P(60_000, 256, tables=True)) == 'lat4'
    # a pinned variant wins over the size rule
    assert fam(P(1000, 256, variant=_lib.KERNEL_VARIANTS['pair'])) == 'pair' and fam(P(10 ** 6, 256, variant=_lib.KERNEL_VARIANTS['throughput'])) == 'thr'
    groups = [(8192, 1), (8193, 2), (16384, 2), (16385, 3), (26624, 3), (26625, 4), (36864, 4), (36865, 2), (53248, 2), (53249, 3), (81920, 3), (81921, 1)]
    for n, g in groups:
        assert P(n, 256)['planning_groups'] == g, (n, g)
    tile32 = [(16384, 1), (26624, 1), (26625, 0), (32768, 0), (32769, 1), (43008, 1), (43009, 0), (100_000, 0)]
    for n, t in tile32:
        assert P(n, 256)['actor_tile32'] == t, (n, t)
    lat = _lib.KERNEL_VARIANTS['latency']   # (the automatic choice in this range is the dual family, below)
    assert P(40_000, 256, tables=True)['combat_latency'] == 1 and P(40_001, 256, tables=True)['combat_latency'] == 0
    # SingleCombat, up to one 128-aircraft tile per CU: the dual8 variant (automatic choice, Euler, MLP numerics only)
    cl = lambda n, **kw: P(n, 256, **kw)['combat_latency']
    assert cl(2) == 1 and cl(16_384) == 1 and cl(16_385) == 2 and cl(32_768) == 2 and cl(32_769) == 3 and cl(65_536) == 3 and cl(65_537) == 0
    assert cl(25_000, tables=True) == 1 and cl(25_000, variant=_lib.KERNEL_VARIANTS['latency']) == 1 and cl(25_000, solver=1) == 1
    # ---- partitions: the same decision at the same rows per CU
    for cus in (32, 128):
        for n, f in expect:
            if n <= 1:
                continue
            m = (n - 1) * cus // 256 + 1 if n % 2 else n * cus // 256   # n = k * 256 x and k * 256 x + 1 map to k * cus x (+ 1)
            assert fam(P(m, cus)) == fam(P(n, 256)), (cus, n, m)
        for n, g in groups:
            m = (n - 1) * cus // 256 + 1 if n % 2 else n * cus // 256
            assert P(m, cus)['planning_groups'] == g, (cus, n, m)
        assert P(5000 * cus // 32, cus, tables=True)['combat_latency'] == 1 and P(5000 * cus // 32 + 1, cus, tables=True)['combat_latency'] == 0
        assert P(64 * cus, cus)['combat_latency'] == 1 and P(64 * cus + 1, cus)['combat_latency'] == 2 and P(128 * cus, cus)['combat_latency'] == 2
        assert P(128 * cus + 1, cus)['combat_latency'] == 3 and P(256 * cus, cus)['combat_latency'] == 3 and P(256 * cus + 1, cus)['combat_latency'] == 0
    # PlanningEnv's inner loop, automatic mode: by 32-row tiles per CU (2 = one persistent workgroup per tile, 4 = guest schedule, 5 = dual
    # workgroups, 1 = launch by launch)
    for cus in (32, 128, 256):
        per = 32 * cus   # aircraft at one tile per CU
        for n, mode in [(1, 2), (per, 2), (per + 1, 4), (per * 3 // 2, 4), (per * 3 // 2 + 32, 5), (2 * per, 5), (2 * per + 1, 1), (40 * per, 1)]:
            assert P(n, cus)['planning_mode'] == mode, (cus, n, mode, P(n, cus))
    assert P(10_000, 256)['planning_mode'] == 4 and P(8_192, 256)['planning_mode'] == 2 and P(16_384, 256)['planning_mode'] == 5
    # block-fixed-point controller: the same up to 1.5 tiles per CU, then the queue (3) up to 1.75 and the guest schedule (4) up to two
    assert [P(n, 256)['planning_mode_i8'] for n in (8_192, 10_000, 12_288, 12_320, 14_336, 14_337, 16_384, 16_385)] == [2, 4, 4, 3, 3, 4, 4, 1]
    # a 32-CU partition runs 1e6 aircraft on the three-wave pair build, like the full device, and 2 048 aircraft still on eight waves per tile
    assert fam(P(1_000_000, 32)) == 'pair3' and fam(P(2048, 32)) == 'lat8' and fam(P(2049, 32)) == 'lat4'
    with pytest.raises(RuntimeError):
        P(0, 256)
    with pytest.raises(RuntimeError):
        P(100, 0)
