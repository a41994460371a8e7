#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh <lib1.so> <lib2.so> ...  (paths relative to the repo)
cd "$(dirname "$0")/.."
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB="$PWD/$lib"; fi
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stages_ms']
print('$lib', d['ms_per_step'], 'fwd', s['composite_fwd'], 'bwd', s['composite_bwd'], 'pre', s['preprocess'], 'scat', s['scatter'], 'sort', s['tile_sort'])"
  done
done
