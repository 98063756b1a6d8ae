#!/usr/bin/env python3
"""Audit the gfx950 ISA of akz_kernels.hip for fused multiply-adds.

Parity with the reference is bit-exact only if no f32/f64 mul+add pair of the image
arithmetic is contracted.  FMAs are legitimate only inside the compiler's correctly rounded
division / sqrt expansions (v_div_scale/v_div_fmas/v_div_fixup, v_rcp/v_rsq refinement), so a
kernel with FMAs but no division/sqrt is flagged.  Prints one line per kernel and exits 1 on a
violation.  Usage: isa_audit.py path/to/akz_kernels.s
"""
import re
import sys


def audit(path):
    s = open(path).read()
    rows, bad = [], []
    for m in re.finditer(r'^(_ZN3akz\w+):[^\n]*\n(.*?)^\.Lfunc_end', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        fma = re.findall(r'\bv_(?:fma|fmac|mad|pk_fma|mac|fmaak|fmamk)_(?:f32|f64|legacy_f32)\b', body)
        fma += re.findall(r'\bv_(?:fmaak|fmamk|madak|madmk|pk_fma)_f32\b', body)
        div = len(re.findall(r'v_div_(?:scale|fmas|fixup)_f(?:32|64)|v_rsq_f(?:32|64)|v_rcp_f(?:32|64)', body))
        short = re.sub(r'^_ZN3akz12_GLOBAL__N_1\d+', '', name)
        rows.append((short, len(fma), div))
        if fma and not div:
            bad.append(short)
    return rows, bad


if __name__ == '__main__':
    rows, bad = audit(sys.argv[1])
    for r in rows:
        print(f"{r[0][:48]:48s} fma={r[1]:3d} div/sqrt_expansion_ops={r[2]:3d}")
    if bad:
        print("FMA without a division/sqrt expansion in:", bad)
        sys.exit(1)
