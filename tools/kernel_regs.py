#!/usr/bin/env python3
"""Register / scratch budget of every kernel of a hipcc -save-temps assembly file:
  hipcc ... -save-temps=obj -c x.hip -o /tmp/x.o ; tools/kernel_regs.py /tmp/x-hip-amdgcn-amd-amdhsa-gfx950.s [filter]
Prints kernel (demangled head), VGPRs, AGPRs, bytes of scratch, and the waves per SIMD the unified 512-register file allows."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
vals = {}
for m in re.finditer(r"\.set (\S+)\.(num_vgpr|num_agpr|private_seg_size), (\d+)", text):
    vals.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
for name, v in vals.items():
    if "num_vgpr" not in v or flt not in name:
        continue
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    vg, ag = v["num_vgpr"], v.get("num_agpr", 0)
    tot = ((vg + 3) // 4 * 4 if ag else vg) + ag
    tot8 = (tot + 7) // 8 * 8
    print(f"{dem[:90]:90s} vgpr {vg:3d} agpr {ag:3d} scratch {v.get('private_seg_size', 0):4d} B  waves/SIMD {min(8, 512 // max(tot8, 1))}")
