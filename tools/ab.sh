#!/bin/bash
# A/B on the GPU box: tools/ab.sh [bench args ...]   (GSR_AB_LIBS="lib1.so lib2.so" to compare builds)
cd "$(dirname "$0")/.."
for lib in default $GSR_AB_LIBS; do
  if [ "$lib" = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB="$PWD/$lib"; fi
  for rep in 1 2; do
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-other-lists "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['roofline']['stages_ms']
print('$lib', d['ms_per_step'], 'D', d['config']['tile_instances'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"
  done
done
