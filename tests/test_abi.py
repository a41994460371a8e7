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
    hdr = open(L.HEADER_PATH).read()
    declared = sorted(set(re.findall(r"GSR_API\s+[\w\s\*]+?\b(gsr_\w+)\s*\(", hdr)))
    assert declared, "no GSR_API declarations parsed"
    assert sorted(L.EXPORTS) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.gsr_version()


def test_stale_callers_fail_loudly(pkg):
    """ADVICE r2 (gsr.h:77): flag bit 1u changed meaning between ABI 1 and 2, structs grew — a caller built against an
    old header must get an error, not the other list mode or mis-sized structs.  Bit 1u is retired and rejected; the
    reference-lists flag lives on bit 2u; gsr_check_abi compares the ABI number and the six struct sizes."""
    L = pkg._lib
    lib = L.load()
    assert lib.gsr_abi_version() == L.ABI_VERSION == 3
    assert b"abi 3" in lib.gsr_version()
    hdr = open(L.HEADER_PATH).read()
    assert re.search(r"#define\s+GSR_ABI_VERSION\s+3\b", hdr) and re.search(r"#define\s+GSR_FLAG_REFERENCE_TILE_LISTS\s+2u", hdr)
    h = C.c_void_p()
    cfg = L.Config(64, 48, 3, 0.2, 1000.0, 3, 0.3, 1, 0)  # the retired bit
    assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG
    assert b"retired" in lib.gsr_last_error_string()
    sizes = [C.sizeof(t) for t in (L.Config, L.Inputs, L.CameraS, L.Aux, L.Stats, L.Grads)]
    assert lib.gsr_check_abi(3, *sizes) == 0
    assert lib.gsr_check_abi(2, *sizes) == L.GSR_E_INVALID_ARG and b"ABI 2" in lib.gsr_last_error_string()
    stale = list(sizes); stale[5] -= 16  # round-1 gsr_grads had no vmeans2d / forward_generation
    assert lib.gsr_check_abi(3, *stale) == L.GSR_E_INVALID_ARG and b"gsr_grads" in lib.gsr_last_error_string()
    # the Julia binding carries the same numbers
    jl = open(os.path.join(os.path.dirname(L.HEADER_PATH), "..", "julia", "GaussianSplattingHipNative.jl")).read()
    assert "const GSR_ABI_VERSION = 3" in jl and "reference_tile_lists ? 0x2 : 0x0" in jl and ":gsr_check_abi" in jl


def test_struct_sizes_match_header(pkg):
    L = pkg._lib
    # include/gsr.h layouts (x86-64 SysV): catches a drifting binding
    assert C.sizeof(L.Config) == 40
    assert C.sizeof(L.Inputs) == 16 + 5 * 8 + 12 + 4
    assert C.sizeof(L.CameraS) == (9 + 3 + 2 + 2 + 3) * 4 + 4 + 16
    assert C.sizeof(L.Stats) == 40
    assert C.sizeof(L.Grads) == 80
    assert C.sizeof(L.Aux) == 24
    assert C.sizeof(L.TailState) == 248


def test_struct_sizes_match_the_c_compiler(pkg, tmp_path):
    """sizeof of every struct of include/gsr.h as gcc lays it out == the ctypes mirror."""
    import subprocess
    L = pkg._lib
    pairs = {"gsr_config": L.Config, "gsr_inputs": L.Inputs, "gsr_camera": L.CameraS, "gsr_aux": L.Aux,
             "gsr_stats": L.Stats, "gsr_grads": L.Grads, "gsr_adam_group": L.AdamGroup, "gsr_tail_grads": L.TailGrads,
             "gsr_tail_state": L.TailState, "gsr_compose_group": L.ComposeGroup, "gsr_gather_group": L.GatherGroup}
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
    cfg = L.Config(64, 48, 4, 0.2, 1000.0, 3, 0.3, 0, 0)  # mode 4 does not exist
    assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG
    assert b"Invalid render mode" in lib.gsr_last_error_string()
    cfg = L.Config(0, 48, 3, 0.2, 1000.0, 3, 0.3, 0, 0)
    assert lib.gsr_create(C.byref(cfg), C.byref(h)) == L.GSR_E_INVALID_ARG
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
