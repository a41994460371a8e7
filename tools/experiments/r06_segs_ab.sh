#!/bin/bash
# on the GPU box: segments per split list in the backward (GSR_BWD_LONG_SEGS = 32 default, 16, 8)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r06; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --no-scenes --steps 20 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' '.join(f'{k}={v:.4f}' for k,v in s.items()))
PY
}
for rep in 1 2; do
for lib in default tools/bin/libgsr_segs16.so tools/bin/libgsr_segs8.so; do
  if [ "$lib" = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/$lib; fi
  run "trained1m $lib" $B --scene trained --seed 1010 --mode rgbd
  run "hot6k     $lib" $B --skew hot:6000 --seed 1003 --no-loss
  run "hot32k    $lib" $B --skew hot:32000 --seed 1003 --no-loss
done
done
for lib in default tools/bin/libgsr_segs16.so tools/bin/libgsr_segs8.so default tools/bin/libgsr_segs16.so tools/bin/libgsr_segs8.so; do
  if [ "$lib" = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/$lib; fi
  python - "$lib" <<'PY'
import os, sys
sys.path.insert(0, "tools")
import gsr_pkg, train_harness as TH
pkg = gsr_pkg.load()
r = TH.protocol_run(pkg, TH.Protocol(densify_grad_threshold=4e-5), warmup=500, steps=1000)
r = r[0] if isinstance(r, tuple) else r
m = r["ms_per_step"]
print("protocol", sys.argv[1], "mean", m["mean"], "median", m["median"], "rounds", r["per_round_median_ms"])
PY
done
