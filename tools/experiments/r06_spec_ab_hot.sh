#!/bin/bash
# on the GPU box: speculative mid-tier sorts on / off on scenes whose views also hold lists beyond 8192
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
for on in 1 0; do
  export GSR_SPEC_TIER_SORTS=$on
  run "hot32k  spec=$on" $B --skew hot:32000 --seed 1003 --no-loss
  run "dense4k spec=$on" $B --skew dense:0.01:50 --seed 1005 --no-loss --gaussians 5000000 --width 3840 --height 2160 --steps 10
done
done
