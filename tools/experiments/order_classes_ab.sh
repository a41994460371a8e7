#!/bin/bash
# On the GPU box: launch-order length classes (GSR_ORDER_CLASSES) vs HBM reads of the fused forward and kernel times.
# the knob GSR_ORDER_CLASSES only exists in profiles/r03/experiments/tile_order_classes.patch — without that patch
# applied and the variant library built, every "class" below would measure the same binary.  Refuse to run then.
grep -q GSR_ORDER_CLASSES gaussiansplatting.jl_amd/csrc/*.hip gaussiansplatting.jl_amd/csrc/*.cpp 2>/dev/null || {
  echo "order_classes_ab.sh: GSR_ORDER_CLASSES is not in this tree: apply profiles/r03/experiments/tile_order_classes.patch and rebuild first" >&2; exit 2; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in "$@"; do
  export GSR_ORDER_CLASSES=$c
  tools/measure_pmc.sh oc$c > /dev/null 2>&1
  python3 - $c <<'PY'
import json, sys
d = json.load(open("profiles/pmc_traffic.json"))["configs"]["N1000000_1920x1080_SH3_rgb_cull_loss"]
print(f"classes {sys.argv[1]:>5s}: sort_composite_fwd HBM read {d['hbm_read']['sort_composite_fwd']/1e6:6.1f} MB, write {d['hbm_write']['sort_composite_fwd']/1e6:6.1f} MB; composite_bwd read {d['hbm_read']['composite_bwd']/1e6:6.1f} MB")
PY
  for rep in 1 2; do tools/kernel_times.sh 2>&1 | grep -E "sort_comp|composite_bwd_kernel|tile_scan" | tr '\n' ' '; echo; done
done
