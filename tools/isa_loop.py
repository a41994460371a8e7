#!/usr/bin/env python3
"""Print the basic block(s) of a kernel in an ISA listing (.s from hipcc -S --cuda-device-only) that contain a given
opcode — e.g. the per-visit body of a compositing kernel (the block with v_exp_f32) — with its VALU cost (tools/isa_cost.py
weights).   tools/isa_loop.py file.s <kernel-name-substring> [opcode=v_exp_f32]"""
import re
import sys

COST = [(r"v_permlane(16|32)_swap", 8.2), (r"v_(exp|rcp|rsq|log|sqrt)_f32", 8.2), (r"v_pk_", 4.3), (r"_dpp", 4.2), (r"v_readlane|v_readfirstlane", 4.2),
        (r"v_cmp", 4.2), (r"v_cndmask", 4.2), (r"v_(fma|mad|fmac|med3|min3|max3)_", 3.6), (r"v_", 2.34)]


def cost(line):
    for pat, c in COST:
        if re.search(pat, line):
            return c
    return 0.0


def blocks(path, kernel):
    s = open(path).read()
    m = re.search(r"^(\S*%s\S*):" % re.escape(kernel), s, re.M)
    body = s[m.end():s.index(".Lfunc_end", m.end())]
    cur, name = [], "entry"
    for raw in body.split("\n"):
        t = raw.strip()
        if not t or t.startswith(";"):
            continue
        lab = re.match(r"^(\.LBB\d+_\d+):", t)
        if lab:
            yield name, cur
            cur, name = [], lab.group(1)
            continue
        if t.startswith("."):
            continue
        cur.append(t.split(";")[0].strip())
        if t.startswith(("s_cbranch", "s_branch")):
            yield name, cur
            cur, name = [], name + "+"
    yield name, cur


if __name__ == "__main__":
    path, kernel = sys.argv[1], sys.argv[2]
    op = sys.argv[3] if len(sys.argv) > 3 else "v_exp_f32"
    for name, b in blocks(path, kernel):
        if any(l.startswith(op) for l in b):
            valu = [l for l in b if l.startswith("v_")]
            print(f"== block {name}: {len(b)} instructions, {len(valu)} VALU, {sum(cost(l) for l in valu):.0f} VALU cycles, "
                  f"{sum(l.startswith('ds_') for l in b)} LDS, {sum(l.startswith('s_') for l in b)} SALU")
            print("\n".join("   " + l for l in b))
