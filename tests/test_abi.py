"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads,
and exports every symbol include/gsr.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest


def test_library_exports_every_declared_symbol(pkg):
    L = pkg._lib
    if not os.path.exists(L.LIB_PATH):
        L.build()
    lib = L.load()
    inc = os.path.dirname(L.HEADER_PATH)
    hdr = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))  # gsr.h + gsr_policy.h
    declared = sorted(set(re.findall(r"GSR_API\s+[\w\s\*]+?\b(gsr_\w+)\s*\(", hdr)))
    assert declared, "no GSR_API declarations parsed"
    assert sorted(L.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.gsr_version()


def test_stale_callers_fail_loudly(pkg):
    """ADVICE r2 (gsr.h:77): flag bit 1u changed meaning between ABI 1 and 2, structs grew — a caller built against an
    old header must get an error, not the other list mode or mis-sized structs.  Bit 1u is retired and rejected; the
    reference-lists flag lives on bit 2u; gsr_check_abi compares the ABI number and the seven struct sizes (ABI 6: + gsr_tail_state,
    which grew late in ABI 5 without being size-checked — ADVICE r5)."""
    L = pkg._lib
    lib = L.load()
    assert lib.gsr_abi_version() == L.ABI_VERSION == 6
    assert b"abi 6" in lib.gsr_version()
    hdr = open(L.HEADER_PATH).read()
    assert re.search(r"#define\s+GSR_ABI_VERSION\s+6\b", hdr) and re.search(r"#define\s+GSR_FLAG_REFERENCE_TILE_LISTS\s+2u", hdr)
    h = C.c_void_p()
    cfg = L.Config(64, 48, 3, 0.2, 1000.0, 3, 0.3, 1, 0, 0, 0, 0, 0)  # the retired bit
    assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG
    assert b"retired" in lib.gsr_last_error_string()
    sizes = [C.sizeof(t) for t in (L.Config, L.Inputs, L.CameraS, L.Aux, L.Stats, L.Grads, L.TailState)]
    assert lib.gsr_check_abi(6, *sizes) == 0
    assert lib.gsr_check_abi(5, *sizes) == L.GSR_E_INVALID_ARG and b"ABI 5" in lib.gsr_last_error_string()
    assert lib.gsr_check_abi(4, *sizes) == L.GSR_E_INVALID_ARG and b"ABI 4" in lib.gsr_last_error_string()
    assert lib.gsr_check_abi(3, *sizes) == L.GSR_E_INVALID_ARG and b"ABI 3" in lib.gsr_last_error_string()
    assert lib.gsr_check_abi(2, *sizes) == L.GSR_E_INVALID_ARG and b"ABI 2" in lib.gsr_last_error_string()
    stale = list(sizes); stale[5] -= 16  # round-1 gsr_grads had no vmeans2d / forward_generation
    assert lib.gsr_check_abi(6, *stale) == L.GSR_E_INVALID_ARG and b"gsr_grads" in lib.gsr_last_error_string()
    # the Julia binding carries the same numbers
    jl = open(os.path.join(os.path.dirname(L.HEADER_PATH), "..", "julia", "GaussianSplattingHipNative.jl")).read()
    stale = list(sizes); stale[3] -= 8  # ABI 3's gsr_aux had no flags / reserved
    assert lib.gsr_check_abi(6, *stale) == L.GSR_E_INVALID_ARG and b"gsr_aux" in lib.gsr_last_error_string()
    stale = list(sizes); stale[0] -= 8  # ABI 5's gsr_config had no form_tuner / grad_precision
    assert lib.gsr_check_abi(6, *stale) == L.GSR_E_INVALID_ARG and b"gsr_config" in lib.gsr_last_error_string()
    stale = list(sizes); stale[6] -= 8  # early ABI 5's gsr_tail_state had no flags / reserved: the library would read 8 bytes past it
    assert lib.gsr_check_abi(6, *stale) == L.GSR_E_INVALID_ARG and b"gsr_tail_state" in lib.gsr_last_error_string()
    assert "const GSR_ABI_VERSION = 6" in jl and "reference_tile_lists ? 0x2 : 0x0" in jl and ":gsr_check_abi" in jl
    assert "sizeof(GsrTailState)" in jl


def test_struct_sizes_match_header(pkg):
    L = pkg._lib
    # include/gsr.h layouts (x86-64 SysV): catches a drifting binding
    assert C.sizeof(L.Config) == 56  # ABI 6: + form_tuner, grad_precision
    assert C.sizeof(L.Inputs) == 16 + 5 * 8 + 12 + 4
    assert C.sizeof(L.CameraS) == (9 + 3 + 2 + 2 + 3) * 4 + 4 + 16
    assert C.sizeof(L.Stats) == 96   # ABI 6: + the view-history block and the tier-tile counts
    assert C.sizeof(L.Grads) == 88   # + flags, reserved (ABI 5, late: GSR_GRADS_COLOR_COTANGENT)
    assert C.sizeof(L.Aux) == 32
    assert C.sizeof(L.TailState) == 256


def test_struct_sizes_match_the_c_compiler(pkg, tmp_path):
    """sizeof of every struct of include/gsr.h as gcc lays it out == the ctypes mirror."""
    import subprocess
    L = pkg._lib
    pairs = {"gsr_config": L.Config, "gsr_inputs": L.Inputs, "gsr_camera": L.CameraS, "gsr_aux": L.Aux,
             "gsr_stats": L.Stats, "gsr_grads": L.Grads, "gsr_adam_group": L.AdamGroup, "gsr_tail_grads": L.TailGrads,
             "gsr_tail_state": L.TailState, "gsr_compose_group": L.ComposeGroup, "gsr_gather_group": L.GatherGroup,
             "gsr_form_tuner": L.FormTuner, "gsr_policy_config": L.PolicyConfig, "gsr_policy_state": L.PolicyState,
             "gsr_view_plan": L.ViewPlan, "gsr_view_outcome": L.ViewOutcome, "gsr_bwd_split": L.BwdSplit}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "gsr.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in pairs) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.dirname(L.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, t in pairs.items():
        assert int(out[n]) == C.sizeof(t), n


def test_invalid_arguments_fail_before_touching_the_gpu(pkg):
    """Error behaviour of the reference: error("Invalid render mode") (rasterizer.jl:51,68)."""
    L = pkg._lib
    lib = L.load()
    h = C.c_void_p()
    cfg = L.Config(64, 48, 4, 0.2, 1000.0, 3, 0.3, 0, 0, 0, 0, 0, 0)  # mode 4 does not exist
    assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG
    assert b"Invalid render mode" in lib.gsr_last_error_string()
    cfg = L.Config(0, 48, 3, 0.2, 1000.0, 3, 0.3, 0, 0, 0, 0, 0, 0)
    assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG
    # the per-handle switches take GSR_DEFAULT (0 since ABI 6: a zero-initialised gsr_config IS the default config)
    # or their explicit values 1 / 2 (grad_precision: 1); ABI 5's -1 is rejected, not read as something else
    assert (L.DEFAULT, L.SSIM_FAST, L.SSIM_EXACT, L.PREPROCESS_DIRECT, L.PREPROCESS_AGGREGATING) == (0, 1, 2, 1, 2)
    for sp, pf, ft, gp, word in ((3, 0, 0, 0, b"ssim_precision"), (-1, 0, 0, 0, b"ssim_precision"), (0, 3, 0, 0, b"preprocess_form"),
                                 (1, -1, 0, 0, b"preprocess_form"), (0, 0, 3, 0, b"form_tuner"), (0, 0, -1, 0, b"form_tuner"),
                                 (0, 0, 0, 3, b"grad_precision"), (0, 0, 0, -1, b"grad_precision")):
        cfg = L.Config(64, 48, 3, 0.2, 1000.0, 3, 0.3, 0, 0, sp, pf, ft, gp)
        assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG and word in lib.gsr_last_error_string()
    # the process-wide defaults: getters return what the setters stored; out-of-range values are rejected and change nothing
    was = (lib.gsr_get_ssim_precision(), lib.gsr_get_preprocess_form())
    try:
        assert lib.gsr_ssim_precision(1) == 0 and lib.gsr_get_ssim_precision() == 1
        assert lib.gsr_ssim_precision(2) == L.GSR_E_INVALID_ARG and lib.gsr_get_ssim_precision() == 1
        assert lib.gsr_preprocess_form(0) == 0 and lib.gsr_get_preprocess_form() == 0
        assert lib.gsr_preprocess_form(-1) == 0 and lib.gsr_get_preprocess_form() == -1
        assert lib.gsr_host_wait_policy(1 << 21, 0, 0) == L.GSR_E_INVALID_ARG and lib.gsr_host_wait_policy(30, 0, 0) == 0
    finally:
        lib.gsr_ssim_precision(was[0]); lib.gsr_preprocess_form(was[1])
    assert lib.gsr_forward(None, None, None, None, None, None, None) == L.GSR_E_INVALID_ARG
    assert lib.gsr_backward(None, None, None, None, None, None) == L.GSR_E_INVALID_ARG
    with pytest.raises(L.GsrError):
        L.check(L.GSR_E_INVALID_ARG)


def test_host_mirror_rejects_cpu_tensors(pkg):
    import torch
    with pytest.raises(ValueError):
        pkg.rasterizer.GaussianRasterizer(64, 48, mode="rgb", device="cpu")
    with pytest.raises(ValueError):
        pkg.rasterizer.n_color_features("rgba")
    with pytest.raises(ValueError):
        pkg.fused_ssim._fused_ssim(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 16, 16))


def test_julia_struct_layouts_match_the_header(pkg, tmp_path):
    """No Julia in the image has ever parsed julia/GaussianSplattingHipNative.jl (round-2 verdict: "type-level risks by
    inspection").  What CAN be checked here: every `struct Gsr*` of the binding, laid out by the C rules Julia applies to an
    isbits struct (natural alignment of Int32 / UInt32 / Float32 / Int64 / UInt64 / Ptr / NTuple fields), must have the
    field names, order, offsets and total size gcc gives the struct of the same role in include/gsr.h."""
    import subprocess
    L = pkg._lib
    jl = open(os.path.join(os.path.dirname(L.HEADER_PATH), "..", "julia", "GaussianSplattingHipNative.jl")).read()
    prim = {"Int32": (4, 4), "UInt32": (4, 4), "Float32": (4, 4), "Int64": (8, 8), "UInt64": (8, 8), "UInt8": (1, 1)}

    def size_align(t):
        t = t.strip()
        if t.startswith("Ptr{"):
            return 8, 8
        m = re.fullmatch(r"NTuple\{(\d+),\s*(.+)\}", t)
        if m:
            s, a = size_align(m.group(2))
            return int(m.group(1)) * s, a
        return prim[t]

    roles = {"GsrConfig": "gsr_config", "GsrInputs": "gsr_inputs", "GsrCamera": "gsr_camera", "GsrAux": "gsr_aux",
             "GsrStats": "gsr_stats", "GsrGrads": "gsr_grads", "GsrTailGrads": "gsr_tail_grads", "GsrTailState": "gsr_tail_state"}
    julia = {}
    for name in roles:
        m = re.search(r"^struct " + name + r";([^\n]*?); end$", jl, re.M) or \
            re.search(r"^struct " + name + r"\n(.*?)^end", jl, re.S | re.M)
        assert m, name
        body = re.sub(r"#[^\n]*", "", m.group(1))
        fields, off, amax = [], 0, 1
        for decl in re.split(r"[;\n]", body):
            decl = decl.strip()
            if not decl:
                continue
            fname, ftype = decl.split("::")
            s, a = size_align(ftype)
            off = (off + a - 1) // a * a
            fields.append((fname.strip(), off, s))
            off += s
            amax = max(amax, a)
        julia[name] = (fields, (off + amax - 1) // amax * amax)
    # the same from the C compiler
    rename = {"opacities_act": "opacities_act"}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gsr.h"', "int main(void){"]
    for jn, cn in roles.items():
        lines.append(f'printf("{jn} sizeof %zu\\n", sizeof({cn}));')
        for fname, _, _ in julia[jn][0]:
            lines.append(f'printf("{jn} {fname} %zu %zu\\n", offsetof({cn}, {rename.get(fname, fname)}), sizeof((({cn}*)0)->{fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.dirname(L.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    c_fields, c_size = {}, {}
    for line in out:
        p = line.split()
        if len(p) == 3:
            c_size[p[0]] = int(p[2])
        elif len(p) == 4:
            c_fields.setdefault(p[0], []).append((p[1], int(p[2]), int(p[3])))
    for jn in roles:
        assert julia[jn][0] == c_fields[jn], (jn, julia[jn][0], c_fields[jn])
        assert julia[jn][1] == c_size[jn], (jn, julia[jn][1], c_size[jn])
    # and the number of fields: nothing of the C struct is missing at the end of the Julia one
    hdr = open(L.HEADER_PATH).read()
    for jn, cn in roles.items():
        body = re.search(r"typedef struct " + cn + r" \{(.*?)\} " + cn + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_c = sum(len(d.split(",")) for d in body.split(";") if d.strip())
        assert n_c == len(julia[jn][0]), (jn, n_c, len(julia[jn][0]))


def test_julia_ccall_signatures_match_the_header(pkg):
    """Every `ccall((:gsr_*, LIB), ret, (argument types...), ...)` of the Julia binding against the prototype of the same
    name in include/gsr.h: same number of arguments, each of the same machine class (pointer / 32-bit int / 64-bit int /
    size_t / float), same return class."""
    L = pkg._lib
    jl = open(os.path.join(os.path.dirname(L.HEADER_PATH), "..", "julia", "GaussianSplattingHipNative.jl")).read()
    hdr = re.sub(r"/\*.*?\*/", "", open(L.HEADER_PATH).read(), flags=re.S)

    def split_top(s):
        out, depth, cur = [], 0, ""
        for ch in s:
            if ch in "({[":
                depth += 1
            elif ch in ")}]":
                depth -= 1
            if ch == "," and depth == 0:
                out.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            out.append(cur.strip())
        return out

    def jl_class(t):
        t = t.strip()
        if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "UInt32": "i32", "Cuint": "i32", "Cfloat": "f32", "Float32": "f32",
                "Int64": "i64", "UInt64": "i64", "Clonglong": "i64", "Csize_t": "size"}[t]

    def c_class(t):
        t = t.strip()
        if "*" in t or "[" in t:
            return "ptr"
        t = re.sub(r"\b(const|GSR_API)\b", "", t).split()
        base = " ".join(t[:-1]) if len(t) > 1 else t[0]  # drop the parameter name
        return {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "float": "f32", "int64_t": "i64", "uint64_t": "i64",
                "size_t": "size", "void": "void"}[base]

    calls = 0
    for m in re.finditer(r"ccall\(\(:(gsr_\w+), LIB\),\s*(\w+),\s*\(", jl):
        name, ret = m.group(1), m.group(2)
        # the argument-type tuple: balanced parentheses from m.end() - 1
        i, depth = m.end() - 1, 0
        for j in range(i, len(jl)):
            depth += jl[j] == "("
            depth -= jl[j] == ")"
            if depth == 0:
                break
        jargs = [jl_class(a) for a in split_top(jl[i + 1:j]) if a]
        pm = re.search(r"GSR_API\s+([\w\s\*]+?)\b" + name + r"\s*\((.*?)\)\s*;", hdr, re.S)
        assert pm, f"{name} is not declared in gsr.h"
        cargs = [] if pm.group(2).strip() == "void" else [c_class(a) for a in split_top(pm.group(2))]
        assert jargs == cargs, (name, jargs, cargs)
        cret = "ptr" if "*" in pm.group(1) else c_class(pm.group(1) + " x")
        assert jl_class(ret) == cret, (name, ret, pm.group(1))
        calls += 1
    assert calls >= 10


def test_julia_files_are_block_and_bracket_balanced(pkg):
    """The little that can be said about Julia syntax without a Julia: in julia/*.jl, with strings and comments removed,
    every bracket closes in order, and at bracket depth 0 (comprehension `for`s and `a[end]` live inside brackets) the
    block openers (module, function, struct, if, for, while, let, begin, do, try, quote, macro) equal the `end`s."""
    root = os.path.join(os.path.dirname(pkg._lib.HEADER_PATH), "..", "julia")
    files = [f for f in os.listdir(root) if f.endswith(".jl")]
    assert files
    for f in files:
        s = open(os.path.join(root, f)).read()
        s = re.sub(r'"""(.*?)"""', '""', s, flags=re.S)
        s = re.sub(r'"(\\.|[^"\\])*"', '""', s)
        s = re.sub(r"#=.*?=#", "", s, flags=re.S)
        s = re.sub(r"#[^\n]*", "", s)
        stack, flat = [], []
        pairs = {")": "(", "]": "[", "}": "{"}
        for ch in s:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and stack.pop() == pairs[ch], (f, "bracket mismatch")
            elif not stack:
                flat.append(ch)
        assert not stack, (f, "unclosed bracket")
        toks = re.findall(r"(?<![\w:.!])(module|function|struct|if|for|while|let|begin|do|try|quote|macro|end)(?![\w!])", "".join(flat))
        opens, ends = sum(t != "end" for t in toks), sum(t == "end" for t in toks)
        assert opens == ends, (f, opens, ends)


def test_bins_capacity_policy(pkg):
    """gsr_bins_capacity_after: the host logic that sizes the fixed-capacity key bins (no GPU): longest list + 25 % where the budget
    holds it; bins for the typical tile (4 x the mean list, >= 1024 keys) where a few lists are far longer — those are scattered a
    second time —; no bins (compact mode) under a budget below twice the mean list; grow-only."""
    f = pkg._lib.load().gsr_bins_capacity_after
    T = 120 * 68
    r64 = lambda v: (v + v // 4 + 63) & ~63  # noqa: E731
    # config 3: 3.9 M instances, longest list 569 -> 768 keys, 50 MB of bins
    assert f(3_888_089, 569, 1920, 1080, 0, 0) == r64(569) == 768
    # grow-only, and never below what is in place
    assert f(3_888_089, 569, 1920, 1080, 0, 1024) == 1024 and f(3_888_089, 1500, 1920, 1080, 0, 1024) == r64(1500)
    # the trained-like 3 M / 1440p scene: longest 2 395 = 6.7 x the mean list fits the default budget (20 x the mean)
    assert f(5_171_694, 2395, 2560, 1440, 0, 0) == r64(2395)
    # one tile of 32 451 instances on config 3's scene: bins for it would take (T + 1) x 40 576 x 8 = 2.6 GB > 645 MB;
    # the bins are sized for the other tiles: max(1024, 4 x 495) + 25 %
    cap = f(4_032_599, 32_451, 1920, 1080, 0, 0)
    assert cap == r64(4 * (4_032_599 // T + 1)) and (T + 1) * cap * 8 < 200 * 2 ** 20 and cap >= 1024
    # ... the same under an explicit budget that allows 1 536 keys per tile
    assert f(4_032_599, 32_451, 1920, 1080, (T + 1) * 8 * 1536, 0) == 1536
    # budgets that hold no useful bins: below twice the mean list, or below 64 keys
    assert f(4_032_599, 32_451, 1920, 1080, (T + 1) * 8 * 512, 0) == 0
    assert f(3_888_089, 569, 1920, 1080, 1, 0) == 0 and f(100, 3, 64, 64, 16 * 8 * 17, 0) == 0
    # a tiny scene: a handful of keys still gets its 64-key bins; nothing rendered: 64 too (the minimum)
    assert f(100, 3, 64, 64, 0, 0) == 64 and f(0, 0, 64, 64, 0, 0) == 64
    # invalid arguments
    assert f(-1, 3, 64, 64, 0, 0) == 0 and f(10, 3, 0, 64, 0, 0) == 0
