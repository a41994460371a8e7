#!/usr/bin/env python3
"""Rough VALU issue cost of an ISA listing (lines from stdin), weighted by the
measured per-instruction costs of tools/valu_rates.hip (profiles/r01_valu_rates.txt)."""
import re, sys, collections
COST = [(r"v_permlane(16|32)_swap", 8.2), (r"v_(exp|rcp|rsq|log|sqrt)_f32", 8.2), (r"v_pk_", 4.9), (r"_dpp", 4.2),
        (r"v_cmp", 4.2), (r"v_cndmask.*s\[", 4.2), (r"v_(fma|mad|med3|min3|max3)_", 3.84), (r"v_.*_e64", 3.84), (r"v_", 2.6)]
tot = 0.0; n = collections.Counter(); cyc = collections.Counter()
for line in sys.stdin:
    t = line.strip()
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    if op.startswith("v_"):
        for pat, c in COST:
            if re.search(pat, t):
                n[pat] += 1; cyc[pat] += c; tot += c; break
    elif op.startswith("s_"): n["salu"] += 1
    elif op.startswith("ds_"): n["lds"] += 1
    else: n[op] += 1
for k, v in n.most_common(): print(f"{k:40s} {v:4d} {cyc.get(k,0):7.1f}")
print("VALU cycles", round(tot, 1))
