#!/bin/bash
# After `gpurun -- bash tools/measure_all.sh <tag>` has merged its files into gpurun_out/: copy what is judged into
# profiles/<tag>/final/ and profiles/pmc_traffic.json.     tools/collect_profiles.sh r04
set -e
cd "$(dirname "$0")/.."
R=${1:-r06}; G=gpurun_out; F=profiles/$R/final
mkdir -p $F
cp $G/$R/bench_driver_line.json $G/$R/kernel_stats.csv $G/$R/bench_2ranks_one_gpu_gloo*.json $F/
for c in cfg2 cfg3 cfg5 cfg3_fwdonly cfg3_rgbd trained_1m_rgbd; do
  mkdir -p $F/pmc_$c
  cp $G/pmc_${R}_$c/*_counter_collection.csv $G/pmc_${R}_$c/parse.log $F/pmc_$c/
done
cp $G/$R/pmc_traffic.json profiles/pmc_traffic.json
python - <<PY
import json
d = json.loads(open("$F/bench_driver_line.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], "ms =", d["value"], d["unit"], "| steady", d["steady_state"]["ms_per_step"],
      "| frac", d["roofline"]["frac"], "| traffic", d["roofline"]["traffic"])
PY
