#!/usr/bin/env python3
"""CPU emulation of the generated two-set phase statements (neuralplane_amd/csrc/np_mlp_asm_dual.inc).

The statements are a few thousand lines of inline asm that no compiler checks for meaning.  This emulator executes their TEXT —
scalar control flow (loop counters, record pointer arithmetic, branches), the weight stream (s_load_dwordx16 into the SGPR
buffers, s_waitcnt), the packed FMAs with their op_sel / op_sel_hi / clamp modifiers, the LDS reads of the inputs and the LDS
writes of the coefficients — on a synthetic weight blob in the KBLOB_DUAL layout, and compares every coefficient of both
accumulator sets with a direct evaluation of the same nets in the order the numerics spec prescribes.  Run twice, with the
scalar loads landing (a) at the s_waitcnt that retires them and (b) at once: if both give the right answers, no instruction reads a
buffer that has a load in flight (the hardware returns scalar loads out of order, anywhere in between).  It also checks that
every load stays inside KBLOB_DUAL.

    python tools/emulate_dual_asm.py                      # check everything, exit status 0 / 1
    python tools/emulate_dual_asm.py --dump x.bin         # + blob, inputs and expected outputs for tools/microbench/nm_harness.hip
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_mlp_asm as G  # noqa: E402

INC = os.path.join(G.CSRC, 'np_mlp_asm_dual.inc')


def make_net(rng, shape):
    IN, H1, H2, H3 = shape
    layers, prev = [], IN
    for h in [H1, H2] + ([H3] if H3 else []):
        layers.append((rng.randn(h, prev).astype(np.float32) * 0.3, rng.randn(h).astype(np.float32) * 0.1))
        prev = h
    return layers, (rng.randn(prev).astype(np.float32) * 0.3, np.float32(rng.randn() * 0.1)), np.float32(1.7), np.float32(-0.3)


def pack_nm(net, shape):
    """one record in the KBLOB_DUAL layout (np_nets.h::dual_record_len)"""
    layers, (w, b), sd, mu = net
    rec = []
    for W, bias in layers:
        out, inn = W.shape
        part = []
        for j in range(out):
            part += [W[j, 0], bias[j]]
        for k in range(1, inn):
            part += list(W[:, k])
        rec += part + [0.0] * (len(part) % 2)
    part = [w[0], b] + list(w[1:])
    rec += part + [0.0] * (len(part) % 2) + [sd, mu]
    return np.array(rec + [0.0] * (G.record_len_nm(*shape) - len(rec)), np.float32)


def spec(net, x):
    """the numerics spec's evaluation order (fp64 arithmetic: the comparison tolerance is 1e-4 relative)"""
    layers, (w, b), sd, mu = net
    h = np.asarray(x, np.float64)
    for W, bias in layers:
        acc = bias.astype(np.float64).copy()
        for k in range(W.shape[1]):
            acc = acc + W[:, k].astype(np.float64) * h[k]
        h = np.clip(acc, 0.0, 1.0)          # ReLU with the clamp's upper bound (activations are carried x 2^-40)
    lo, hi = float(b), 0.0
    for k in range(len(w)):
        if k % 2 == 0:
            lo += float(w[k]) * h[k]
        else:
            hi += float(w[k]) * h[k]
    return (lo + hi) * float(sd) + float(mu)


def build_blob(seed=0):
    rng = np.random.RandomState(seed)
    nets = {}
    blob = np.zeros(G.class_base_nm(len(G.CLASSES)) + 2 * G.GROUP, np.float32)
    for ci, (_, shape, _, count, _) in enumerate(G.CLASSES):
        ln = G.record_len_nm(*shape)
        for m in range(count):
            nets[(ci, m)] = make_net(rng, shape)
            o = G.class_base_nm(ci) + m * ln
            blob[o:o + ln] = pack_nm(nets[(ci, m)], shape)
    return blob, nets, rng.randn(9) * 0.5, rng.randn(9) * 0.5


def statement(src, kind, wave):
    name = f'mlp_phase_asm_dual_{kind}_{wave}'
    i = src.index('void ' + name)
    blk = src[i:src.index('constexpr int MLP_PAIR', i)]
    start = int(re.search(r'MLP_PAIR_%s_%d_START = (\d+)' % (kind, wave), src).group(1))
    return [m.group(1) for m in re.finditer(r'^\s*"(.*?)\\n\\t"', blk, re.M)], start


def run_statement(lines, start, blob, xa, xb, land_at_wait):
    """-> ({(set, slot): value}, [byte offsets of all loads])"""
    labels = {ln[:-1]: k for k, ln in enumerate(lines) if ln.endswith(':')}
    S = {'s100': start * 4, 's101': 0}
    SF = np.zeros(128, np.float32)
    V = np.zeros(256, np.float64)
    VA, writes, pending, loads = {}, {}, [], []
    scc, pc, steps = 0, 0, 0

    def sv(tok):
        tok = tok.strip()
        if tok in ('vcc_lo', 'vcc_hi') or tok.startswith('s'):
            return S.get(tok, 0)
        return int(tok, 0)

    while pc < len(lines):
        ln = lines[pc]
        pc += 1
        steps += 1
        assert steps < 3_000_000, 'runaway'
        if ln.endswith(':'):
            continue
        op, _, rest = ln.partition(' ')
        if op == 's_nop':
            continue
        if op == 's_waitcnt':
            for dst, off in pending:
                SF[dst:dst + 16] = blob[off // 4:off // 4 + 16]
            pending = []
            continue
        a = [t.strip() for t in re.split(r',\s*(?![^\[]*\])', re.split(r' op_sel| clamp| offset', rest)[0])]
        if op == 's_load_dwordx16':
            dst = int(re.match(r's\[(\d+):', a[0]).group(1))
            off = S.get('s' + re.match(r's\[(\d+):', a[1]).group(1), 0) + int(a[2], 0)
            assert 0 <= off and off + 64 <= 4 * len(blob), f'load outside KBLOB_DUAL: {off}'
            loads.append(off)
            if land_at_wait:
                pending.append((dst, off))
            else:
                SF[dst:dst + 16] = blob[off // 4:off // 4 + 16]
        elif op == 's_mov_b64':
            if '%[w]' not in rest:
                d, s_ = (int(re.match(r's\[(\d+):', t).group(1)) for t in a[:2])
                S[f's{d}'], S[f's{d + 1}'] = S.get(f's{s_}', 0), S.get(f's{s_ + 1}', 0)
        elif op == 's_mov_b32':
            S[a[0]] = sv(a[1])
        elif op in ('s_add_u32', 's_addc_u32'):
            S[a[0]] = sv(a[1]) + sv(a[2])
        elif op == 's_sub_u32':
            S[a[0]] = sv(a[1]) - sv(a[2])
        elif op == 's_cmp_eq_u32':
            scc = int(sv(a[0]) == sv(a[1]))
        elif op == 's_cmp_lg_u32':
            scc = int(sv(a[0]) != sv(a[1]))
        elif op == 's_cmov_b32':
            if scc:
                S[a[0]] = sv(a[1])
        elif op == 's_cbranch_scc0':
            if not scc:
                pc = labels[a[0]]
        elif op == 's_cbranch_scc1':
            if scc:
                pc = labels[a[0]]
        elif op == 'ds_read_b32':      # the normalised inputs: column NUM_LIVE + group of the own (addra) or the partner's (addrb) lanes
            col = int(re.search(r'offset:%\[step\]\*(\d+)', rest).group(1))
            V[int(a[0][1:])] = (xa if 'addra' in a[1] else xb)[col - G.NUM_LIVE]
        elif op == 'v_add_u32':        # output column addresses
            vd = int(a[0][1:])
            VA[vd] = (0 if 'addra' in a[2] else 1, int(a[1].split('*')[1])) if a[1].startswith('%[step]*') else (VA[vd][0], VA[vd][1] + 1)
        elif op == 'ds_write_b32':
            writes[VA[int(a[0][1:])]] = np.float32(V[int(a[1][1:])])
        elif op in ('v_pk_fma_f32', 'v_pk_mul_f32', 'v_pk_add_f32'):
            m = re.search(r'op_sel:\[([\d,]+)\]', rest)
            osel = [int(t) for t in m.group(1).split(',')] if m else None
            m = re.search(r'op_sel_hi:\[([\d,]+)\]', rest)
            oselh = [int(t) for t in m.group(1).split(',')] if m else None

            def operand(tok, i, hi):
                sel = oselh if hi else osel
                h = sel[i] if sel else (1 if hi else 0)
                mm = re.match(r's\[(\d+):', tok)
                if mm:
                    return float(SF[int(mm.group(1)) + h])
                mm = re.match(r'v\[(\d+):', tok)
                return V[int(mm.group(1)) + h] if mm else float(tok)
            d = int(re.match(r'v\[(\d+):', a[0]).group(1))
            if op == 'v_pk_fma_f32':
                r = [np.float32(operand(a[1], 0, hi) * operand(a[2], 1, hi) + operand(a[3], 2, hi)) for hi in (0, 1)]
            elif op == 'v_pk_mul_f32':
                r = [np.float32(operand(a[1], 0, hi) * operand(a[2], 1, hi)) for hi in (0, 1)]
            else:
                r = [np.float32(operand(a[1], 0, hi) + operand(a[2], 1, hi)) for hi in (0, 1)]
            if ' clamp' in rest:
                r = [min(max(float(t), 0.0), 1.0) for t in r]
            V[d], V[d + 1] = r
        else:
            raise ValueError('instruction not modelled: ' + ln)
    return writes, loads


def check(land_at_wait, seed=0):
    """-> list of mismatch descriptions (empty = every coefficient of every statement right, every load in range)"""
    src = open(INC).read()
    blob, nets, xa, xb = build_blob(seed)
    ci_of = {c[0]: i for i, c in enumerate(G.CLASSES)}
    bad = []
    for kind, waves in G.PAIR_PLANS.items():
        for wave, items in enumerate(waves):
            lines, start = statement(src, kind, wave)
            w, _ = run_statement(lines, start, blob, xa, xb, land_at_wait)
            for cname, first, n in items:
                ci = ci_of[cname]
                grps = G.CLASSES[ci][2]
                for m in range(first, first + n):
                    for st, x in ((0, xa), (1, xb)):
                        want = spec(nets[(ci, m)], [x[G.G[g]] for g in grps])
                        got = w.get((st, G.class_slot(ci) + m))
                        if got is None or abs(got - want) > 1e-4 * max(1.0, abs(want)):
                            bad.append(f'{kind}_{wave} {cname}[{m}] set {st}: {got} != {want}')
    return bad


def dump(path, seed=0):
    blob, nets, xa, xb = build_blob(seed)
    ci_of = {c[0]: i for i, c in enumerate(G.CLASSES)}
    exp = np.full((2, 42), np.nan, np.float32)
    for items in G.PAIR_PLANS['ALL']:
        for cname, first, n in items:
            ci = ci_of[cname]
            for m in range(first, first + n):
                for st, x in ((0, xa), (1, xb)):
                    exp[st, G.class_slot(ci) + m] = spec(nets[(ci, m)], [x[G.G[g]] for g in G.CLASSES[ci][2]])
    with open(path, 'wb') as f:
        f.write(np.int32(len(blob)).tobytes() + blob.tobytes() + xa.astype(np.float32).tobytes() + xb.astype(np.float32).tobytes() + exp.tobytes())


if __name__ == '__main__':
    problems = check(True) + check(False)
    print('\n'.join(problems[:20]) if problems else 'np_mlp_asm_dual.inc: all phase statements evaluate every net of both sets correctly '
          '(loads landing at the wait and at once), all loads inside KBLOB_DUAL')
    if '--dump' in sys.argv:
        dump(sys.argv[sys.argv.index('--dump') + 1])
    sys.exit(1 if problems else 0)
