#!/usr/bin/env python3
"""Audit a gfx950 .s file: between an inline-asm `s_load_dwordx16` and the next inline-asm
`s_waitcnt lgkmcnt(0)`, no other instruction may touch the destination SGPRs of the load (the
compiler does not know the data lands asynchronously).  Usage: audit_smem_asm.py file.s [kernel-substr]"""
import re
import sys


def sregs(tok):
    out = set()
    for m in re.finditer(r'\bs\[(\d+):(\d+)\]', tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bs(\d+)\b', tok):
        out.add(int(m.group(1)))
    return out


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else None
    lines = open(path).read().split('\n')
    in_asm = False
    outstanding = set()
    kernel = None
    bad = 0
    nload = 0
    for ln, line in enumerate(lines, 1):
        t = line.strip()
        m = re.match(r'^(_Z\w+):', t)
        if m:
            kernel = m.group(1)
            outstanding = set()
        if want and (kernel is None or want not in kernel):
            continue
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
            if t.endswith(':') and outstanding and not t.startswith('.LBB'):
                pass
            continue
        code = t.split(';')[0]
        if in_asm:
            if code.startswith('s_load_dwordx16'):
                dst = code.split()[1].rstrip(',')
                outstanding |= sregs(dst)
                nload += 1
            elif code.startswith('s_waitcnt'):
                outstanding = set()
            continue
        if outstanding:
            ops = code.split(None, 1)
            used = sregs(ops[1]) if len(ops) > 1 else set()
            hit = used & outstanding
            if hit:
                bad += 1
                if bad <= 25:
                    print(f'{kernel} line {ln}: `{code.strip()}` touches in-flight SGPRs {sorted(hit)}')
            if code.startswith(('s_cbranch', 's_branch', 's_endpgm', 's_setpc', 's_swappc')):
                print(f'{kernel} line {ln}: control flow `{code.strip()}` with {len(outstanding)} SGPRs in flight')
                bad += 1
    print(f'{nload} asm loads audited, {bad} violations')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
