#!/usr/bin/env python3
"""Instruction budget of composite_bwd_kernel<3, 4, false> from its ISA (round-2 verdict item 4-i).

Compiles composite.hip to an ISA listing exactly as the Makefile does (hipcc -S --cuda-device-only, same flags), splits the
kernel into basic blocks (tools/isa_loop.py), assigns every block a ROLE from its content and position, multiplies its VALU
instruction count by how often that role executes on the measured view (tools/bwd_exec_counts.py ->
profiles/r03/bwd_exec_counts_cfg3.json) and compares the total with the PMC-counted SQ_INSTS_VALU per launch
(profiles/pmc_traffic.json).  Each block is also priced in VALU issue cycles with the measured per-instruction costs of
profiles/r02/valu_rates_clock.txt (2.34 simple / 3.6 three-source FMA / 4.2 cmp, cndmask, DPP, readlane / 8.2 exp, rcp,
permlane swap), which turns the budget into a time at the clock the chip holds under a VALU stream.

  python tools/isa_cost.py [--asm file.s] [--counts profiles/r03/bwd_exec_counts_cfg3.json] [--json out.json]

Roles (execution count on one view):
  prologue        per tile (workgroup)                         pixel state, tile_last reduction
  stage           per batch of 64 staged instances             stream -> LDS
  ballot          per 64-entry ballot (= per batch here)       row-mask work list
  head            per candidate instance                       3 LDS reads, dx, sigma's dx part, row-mask bits
  test[q]         per visited 16x4 pixel group q               dy, sigma, the two compares (:rgb: live, sigma-bits < threshold)
  active[q]       per visited group with >= 1 active lane      exp, alpha, rcp, T, A, P/U1/U2, colour sums
  skip[q]         per candidate whose mask misses group q      (scalar branch; the q = 0 path zero-fills 7 accumulators)
  reduce          per instance with >= 1 active lane           row-then-column wave64 reduction + LDS row store
  glue            per candidate                                loop control, ballot accumulation
  flush           per batch                                    sum slabs, apply -o/2 and conic factors, 48-byte row store
  tail            per tile                                     zero rows behind tile_last
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_loop import blocks, cost  # noqa: E402

KERNEL = "composite_bwd_kernelILi3ELi4ELb0ELb0E"  # <C = 3, PPL = 4, LISTED = false, BG0 = false>: the kernel the :rgb launch takes
N_SIMD = 1024


def compile_asm():
    out = os.path.join(tempfile.mkdtemp(), "composite.s")
    src = os.path.join(ROOT, "gaussiansplatting.jl_amd", "csrc", "composite.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                    "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    return out


def classify(blist):
    """-> list of (name, instrs, role), from the kernel's control flow as the compiler lays it out (it rotates the batch
    loop: the flush of batch k sits in front of the staging of batch k+1):
      * everything in front of the first block with a global_store_dwordx4 (the flush) is the per-tile prologue;
      * the instance loop runs from the block with s_ff1 + ds_read_b128 (head) to the last block that branches back to it;
        inside it, the blocks that compute sigma, compare and narrow EXEC are the four group tests, the all-VALU block
        with v_rcp behind each is its active part, a block of zero-fills is the q = 0 "not visited" path, v_permlane32_swap marks the reduction (and the
        ds_write block behind it), the rest is glue;
      * the batch loop is everything else between the flush and the loop's exit label (the largest label a block in front
        of the head branches to): flush (global stores), stage (global_load_dwordx4 + ds_write_b128), ballot (the rest);
      * from the exit label on: the per-tile tail (zero rows behind tile_last)."""
    names = [n for n, _ in blist]
    txts = ["\n".join(b) for _, b in blist]
    has = lambda i, pat: re.search(pat, txts[i]) is not None  # noqa: E731
    flush0 = next(i for i in range(len(blist)) if has(i, r"global_store_dwordx4"))
    head = next(i for i in range(len(blist)) if has(i, r"s_ff1_i32_b64") and has(i, r"ds_read_b128"))

    def label_index(lbl):
        return next((i for i, n in enumerate(names) if n == lbl), -1)
    # the instance loop's header is the label closest in front of the head that blocks behind the head branch back to
    # (the `while (wl)` test); its back edges are the `continue` of a splat no lane touches and the end of the body
    back = [(label_index(m), i) for i in range(head, len(blist))
            for m in re.findall(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", txts[i]) if 0 <= label_index(m) <= head]
    header = max(t for t, _ in back)
    loop_end = max(i for t, i in back if header - 3 <= t <= header)  # (the compiler splits the header: .._63 falls into .._64)
    exits = [label_index(m) for i in range(flush0, head) for m in re.findall(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", txts[i])]
    exit_idx = max([e for e in exits if e > loop_end], default=len(blist))
    roles = []
    q = -1
    for i, (name, b) in enumerate(blist):
        if i < flush0:
            role = "prologue"
        elif i >= exit_idx:
            role = "tail"
        elif head <= i <= loop_end:
            if i == head:
                role = "head"
            elif has(i, r"v_rcp_f32") and all(l.startswith("v_") for l in b):
                role = f"active{q}"
            elif has(i, r"v_cmp") and has(i, r"s_and_saveexec_b64") and has(i, r"v_fma"):
                q += 1
                role = f"test{q}"
            elif has(i, r"v_permlane32_swap") or (roles and roles[-1][2] == "reduce" and has(i, r"ds_write")):
                role = "reduce"
            elif sum(l.startswith("v_mov_b32") for l in b) >= 5:
                role = f"skip{q + 1}"
            else:
                role = "glue"
        elif has(i, r"global_store_dwordx4") or (has(i, r"ds_read") and has(i, r"v_add_f32") and not has(i, r"global_load")):
            role = "flush"
        elif has(i, r"global_load_dwordx4") and has(i, r"ds_write_b128"):
            role = "stage"
        else:
            role = "ballot"
        roles.append((name, b, role))
    return roles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", default=None)
    ap.add_argument("--counts", default=os.path.join(ROOT, "profiles", "r03", "bwd_exec_counts_cfg3.json"))
    ap.add_argument("--pmc", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    ap.add_argument("--pmc-key", default="N1000000_1920x1080_SH3_rgb_cull_loss")
    ap.add_argument("--json", default=None)
    ap.add_argument("--list", action="store_true", help="print every block with its role")
    a = ap.parse_args()
    asm = a.asm or compile_asm()
    c = json.load(open(a.counts))
    blist = [(n, b) for n, b in blocks(asm, KERNEL) if b]
    roles = classify(blist)
    T, batches, cand = c["tiles_nonempty"], c["batches_of_64"], c["candidates_after_row_mask_ballot"]
    vq, aq = c["group_visits_by_q"], c["active_group_visits_by_q"]
    execs = {"prologue": T, "tail": T, "stage": batches, "ballot": batches, "flush": batches, "head": cand, "glue": cand,
             "reduce": c["instances_reduced"]}
    for q in range(4):
        execs[f"test{q}"] = vq[q]
        execs[f"active{q}"] = aq[q]
        execs[f"skip{q}"] = cand - vq[q]
    agg = {}
    for name, b, role in roles:
        valu = [l for l in b if l.startswith("v_")]
        r = agg.setdefault(role, {"blocks": 0, "instr": 0, "valu": 0, "cycles": 0.0, "lds": 0, "salu": 0})
        r["blocks"] += 1; r["instr"] += len(b); r["valu"] += len(valu); r["cycles"] += sum(cost(l) for l in valu)
        r["lds"] += sum(l.startswith("ds_") for l in b); r["salu"] += sum(l.startswith("s_") for l in b)
        if a.list:
            print(f"{name:16s} {role:10s} {len(b):3d} instr {len(valu):3d} VALU")
    order = ["prologue", "stage", "ballot", "head", "test0", "active0", "skip0", "test1", "active1", "skip1", "test2", "active2",
             "skip2", "test3", "active3", "skip3", "reduce", "glue", "flush", "tail"]
    print(f"{'role':10s} {'blocks':>6s} {'VALU':>5s} {'cycles':>7s} {'LDS':>4s} {'SALU':>5s} {'executions':>11s} {'VALU x exec (M)':>16s} {'cycles x exec (G)':>18s}")
    tot_v = tot_c = 0.0
    rows = []
    for role in order:
        if role not in agg:
            continue
        r = agg[role]; n = execs.get(role, 0)
        tv, tc = r["valu"] * n / 1e6, r["cycles"] * n / 1e9
        tot_v += tv; tot_c += tc
        rows.append({"role": role, **r, "executions": n, "valu_M": round(tv, 2), "cycles_G": round(tc, 3)})
        print(f"{role:10s} {r['blocks']:6d} {r['valu']:5d} {r['cycles']:7.0f} {r['lds']:4d} {r['salu']:5d} {n:11d} {tv:16.2f} {tc:18.3f}")
    measured = None
    try:
        measured = json.load(open(a.pmc))["configs"][a.pmc_key]["sq"]["composite_bwd"]["SQ_INSTS_VALU"]
    except Exception:
        pass
    per_inst = tot_v * 1e6 / cand
    print(f"\nmodel: {tot_v:.1f} M VALU wave-instructions per launch ({per_inst:.1f} per candidate instance), "
          f"{tot_c:.3f} G issue cycles = {tot_c * 1e9 / N_SIMD / 1e6:.3f} M cycles per SIMD "
          f"= {tot_c * 1e9 / N_SIMD / 2.4e9 * 1e3:.3f} ms at 2.4 GHz / {tot_c * 1e9 / N_SIMD / 2.1e9 * 1e3:.3f} ms at 2.1 GHz")
    if measured:
        print(f"PMC:   {measured / 1e6:.1f} M SQ_INSTS_VALU per launch  ->  model / measured = {tot_v * 1e6 / measured:.3f}")
    if a.json:
        json.dump({"kernel": KERNEL, "counts": a.counts, "rows": rows, "model_valu_M": round(tot_v, 2),
                   "model_cycles_G": round(tot_c, 3), "measured_SQ_INSTS_VALU": measured,
                   "model_over_measured": round(tot_v * 1e6 / measured, 4) if measured else None}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
