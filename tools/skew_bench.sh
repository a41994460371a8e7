#!/bin/bash
# On the GPU box: the skewed-scene bench modes: tile_sort time and unsorted-key bytes for one tile with
# 8 k / 32 k / 128 k instances at 1080p and for a 4K view with 1 % of the tiles at 50x the mean density.
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-skew}
mkdir -p $OUT
for k in hot:8000 hot:32000 hot:128000; do
  python bench.py --no-cpu-baseline --no-other-lists --no-loss --steps 5 --warmup 2 --skew $k > $OUT/skew_${k/:/_}.json 2>$OUT/err.log
done
python bench.py --no-cpu-baseline --no-other-lists --no-loss --steps 5 --warmup 2 --gaussians 5000000 --width 3840 --height 2160 --skew dense:0.01:50 > $OUT/skew_4k_dense.json 2>>$OUT/err.log
python - $OUT <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/skew_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "FAILED", e); continue
    st = d["roofline"]["stages_ms"]; b = d["config"]["binning"]
    print(f.split("/")[-1], "ms/step", d["ms_per_step"], "tile_sort", st.get("tile_sort"), "preprocess", st.get("preprocess"),
          "sort_composite_fwd", st.get("sort_composite_fwd"), "composite_fwd", st.get("composite_fwd"), "composite_bwd", st.get("composite_bwd"), b)
PY
