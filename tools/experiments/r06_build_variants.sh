#!/bin/bash
# Round 6: A/B builds of libgsr_hip.so into tools/bin/ (they travel to the GPU box; GSR_HIP_LIB selects one).
#   tools/experiments/r06_build_variants.sh NAME "EXTRA_COMPOSITE flags" ["EXTRA flags for every TU"]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; FLAGS=$2; ALL=$3
W=/tmp/gsr_var/$NAME
rm -rf "$W"; mkdir -p "$W/gaussiansplatting.jl_amd" "$ROOT/tools/bin"
cp -r "$ROOT/include" "$W/include"
mkdir -p "$W/gaussiansplatting.jl_amd/csrc" && cp "$ROOT"/gaussiansplatting.jl_amd/csrc/*.* "$ROOT"/gaussiansplatting.jl_amd/csrc/Makefile "$W/gaussiansplatting.jl_amd/csrc/"
make -C "$W/gaussiansplatting.jl_amd/csrc" -j6 EXTRA_COMPOSITE="$FLAGS" EXTRA="$ALL" OUT="$ROOT/tools/bin/libgsr_$NAME.so" 2>&1 | grep -i "error\|warning" || true
ls -la "$ROOT/tools/bin/libgsr_$NAME.so"
