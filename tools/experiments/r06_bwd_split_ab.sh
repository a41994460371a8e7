#!/bin/bash
# on the GPU box: the backward's list split (tiers of up to GSR_BWD_SPLIT_TILES tiles walked in 32 list segments on the second
# stream) on scenes whose tier tiles are a few hundred lists of 1-2.5 k entries — worth it there?
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
for lim in 256 0 ${EXTRA_LIMS}; do
  export GSR_BWD_SPLIT_TILES=$lim
  run "trained1m split<=$lim" $B --scene trained --seed 1010 --mode rgbd
  run "trained3m split<=$lim" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
  run "hot6k     split<=$lim" $B --skew hot:6000 --seed 1003 --no-loss
  run "hot32k    split<=$lim" $B --skew hot:32000 --seed 1003 --no-loss
done
done
for lim in 256 0 256 0; do
  export GSR_BWD_SPLIT_TILES=$lim
  python - <<'PY'
import os, sys
sys.path.insert(0, "tools")
import gsr_pkg, train_harness as TH
pkg = gsr_pkg.load()
r = TH.protocol_run(pkg, TH.Protocol(densify_grad_threshold=4e-5), warmup=500, steps=1000)
r = r[0] if isinstance(r, tuple) else r
m = r["ms_per_step"]
print("protocol split<=" + os.environ["GSR_BWD_SPLIT_TILES"], "mean", m["mean"], "median", m["median"], "rounds", r["per_round_median_ms"])
PY
done
