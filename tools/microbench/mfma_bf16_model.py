"""Which arithmetic model reproduces v_mfma_f32_16x16x32_bf16 bit for bit?  Reads the dump of tools/microbench/mfma_bf16_model.hip
(one instruction per trial, operands and result as bit patterns) and evaluates candidate models with exact rational arithmetic.
    python tools/microbench/mfma_bf16_model.py gpurun_out/.../mfma_bf16_model.bin"""
import struct, sys
from fractions import Fraction
import numpy as np

def f32_to_frac(x):
    return Fraction(float(x))

def ilog2(a):
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    return e

def round_f32(q, mode='rne', ftz=False):
    """exact rational -> nearest fp32 (as a numpy float32), round-to-nearest-even or toward zero; optional flush of denormal results"""
    if q == 0:
        return np.float32(0.0)
    s = -1 if q < 0 else 1
    a = abs(q)
    # exponent e with 2^e <= a < 2^(e+1)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    e_eff = max(e, -126)
    ulp = Fraction(2) ** (e_eff - 23)
    n = a / ulp
    fl = n.numerator // n.denominator
    rem = n - fl
    if mode == 'rne':
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
    r = fl * ulp
    if ftz and r < Fraction(2) ** -126:
        r = Fraction(0)
    v = float(r)
    if v > 3.4028234663852886e38:
        v = float('inf')
    return np.float32(s * v)

def main(path):
    raw = open(path, 'rb').read()
    trials, kinds = struct.unpack('ii', raw[:8])
    off = 8
    A = np.frombuffer(raw, np.uint16, trials * 512, off).reshape(trials, 16, 32); off += trials * 1024
    B = np.frombuffer(raw, np.uint16, trials * 512, off).reshape(trials, 32, 16); off += trials * 1024
    C = np.frombuffer(raw, np.float32, trials * 256, off).reshape(trials, 16, 16); off += trials * 1024
    D = np.frombuffer(raw, np.float32, trials * 256, off).reshape(trials, 16, 16)
    bf = lambda u: (u.astype(np.uint32) << 16).view(np.float32)
    Af, Bf = bf(A), bf(B)
    contiguous = lambda g: [list(range(s, s + g)) for s in range(0, 32, g)]
    by_e = [[e + 8 * q for q in range(4)] for e in range(8)]
    halves_e = [[k for k in range(32) if k % 8 < 4], [k for k in range(32) if k % 8 >= 4]]
    chunkings = {'all32': [list(range(32))], 'c16': contiguous(16), 'c8': contiguous(8), 'c4': contiguous(4), 'c2': contiguous(2), 'c1': contiguous(1),
                 'by_e(4x8)': by_e, 'halves_by_e': halves_e}
    models = {}
    for cname, ch in chunkings.items():
        for mode in ('rne', 'rtz'):
            for ftz in (False, True):
                models[f'acc_chunks[{cname}],{mode}{",ftz" if ftz else ""}'] = ('acc', ch, mode, ftz)
    for cname in ('all32', 'c16', 'c8', 'halves_by_e'):
        for mode in ('rne', 'rtz'):
            models[f'products_first[{cname}],{mode}'] = ('pf', chunkings[cname], mode, False)
    for W in range(24, 41):
        for tr in ('tz', 'floor'):
            models[f'aligned[c8],W={W},{tr},rne'] = ('al', contiguous(8), W, tr)
    for W in (23, 24, 25):
        models[f'H2: products aligned to max(ea + eb) (grid 2^(emax-{W}), toward zero), exact sum, then RNE(acc + S) [c8]'] = ('h2', contiguous(8), W, 'tz')
    for W in (23, 24, 25):
        models[f'H1: products aligned to their max (W={W}, toward zero), exact sum, then RNE(acc + S) [c8]'] = ('h1', contiguous(8), W, 'tz')
    per_kind = {}
    rows = [(t, i, j) for t in range(trials) for (i, j) in ((0, 0), (3, 7), (9, 12), (15, 15), (6, 1), (12, 5))]
    for (t, i, j) in rows:
        kind = t % kinds
        p = [f32_to_frac(Af[t, i, k]) * f32_to_frac(Bf[t, k, j]) for k in range(32)]
        c = f32_to_frac(C[t, i, j])
        d = D[t, i, j]
        st = per_kind.setdefault(kind, {'n': 0, 'hits': {m: 0 for m in models}})
        st['n'] += 1
        for name, (how, ch, mode, ftz) in models.items():
            if how == 'h2':
                W = mode
                acc = c
                for grp in ch:
                    es = [ilog2(abs(f32_to_frac(Af[t, i, k]))) + ilog2(abs(f32_to_frac(Bf[t, k, j]))) for k in grp if p[k] != 0]
                    if not es:
                        continue
                    grid = Fraction(2) ** (max(es) - W)
                    S = Fraction(0)
                    for k in grp:
                        q = p[k] / grid
                        fl = q.numerator // q.denominator
                        if q < 0 and fl != q:
                            fl += 1
                        S += fl * grid
                    acc = f32_to_frac(round_f32(acc + S, 'rne', False))
                r = np.float32(float(acc))
            elif how == 'h1':
                W = mode
                acc = c
                for grp in ch:
                    nz = [abs(p[k]) for k in grp if p[k] != 0]
                    if not nz:
                        continue
                    grid = Fraction(2) ** (max(ilog2(x) for x in nz) - W)
                    S = Fraction(0)
                    for k in grp:
                        q = p[k] / grid
                        fl = q.numerator // q.denominator
                        if q < 0 and fl != q:
                            fl += 1
                        S += fl * grid
                    acc = f32_to_frac(round_f32(acc + S, 'rne', False))
                r = np.float32(float(acc))
            elif how == 'al':
                W, tr = mode, ftz
                acc = c
                for grp in ch:
                    terms = [acc] + [p[k] for k in grp]
                    nz = [abs(x) for x in terms if x != 0]
                    if not nz:
                        acc = Fraction(0); continue
                    emax = max(ilog2(x) for x in nz)
                    grid = Fraction(2) ** (emax - W)
                    tot = Fraction(0)
                    for x in terms:
                        q = x / grid
                        fl = q.numerator // q.denominator          # floor
                        if tr == 'tz' and q < 0 and fl != q:
                            fl += 1
                        tot += fl * grid
                    acc = f32_to_frac(round_f32(tot, 'rne', False))
                r = np.float32(float(acc))
            elif how == 'acc':
                acc = c
                for grp in ch:
                    acc = f32_to_frac(round_f32(acc + sum(p[k] for k in grp), mode, ftz))
                r = np.float32(float(acc))
            else:
                s = Fraction(0)
                for grp in ch:
                    s = f32_to_frac(round_f32(s + sum(p[k] for k in grp), mode, ftz))
                r = round_f32(c + s, mode, ftz)
            if r.view(np.uint32) == d.view(np.uint32) or (r == 0 and d == 0):
                st['hits'][name] += 1
    names = ['small integers', 'uniform (-1, 1)', 'exponents +-8', 'exponents +-8, |C| ~ 2^10', 'exponents +-20', 'products / C near the denormal range']
    for kind in sorted(per_kind):
        st = per_kind[kind]
        print(f'kind {kind} ({names[kind]}): {st["n"]} elements')
        best = sorted(st['hits'].items(), key=lambda kv: -kv[1])[:8]
        for name, h in best:
            print(f'    {h:4d} / {st["n"]}  {name}')
    total = {m: sum(per_kind[k]['hits'][m] for k in per_kind if k != 5) for m in models}
    n_tot = sum(per_kind[k]['n'] for k in per_kind if k != 5)
    print('over kinds 0-4:')
    for name, h in sorted(total.items(), key=lambda kv: -kv[1])[:10]:
        print(f'    {h:4d} / {n_tot}  {name}')

if __name__ == '__main__':
    main(sys.argv[1])
