#!/bin/bash
# On the GPU box: step time and host CPU time per step of bench.py under different gsr_host_wait_policy settings
# (config 2 = 0.23 ms steps, config 3 = 1.5 ms steps).   tools/experiments/host_wait_ab.sh
cd "$GRAFT_REPO_ROOT"
for hw in default 1000000,0,0 default 20,0,20 300,0,50; do
  for cfg in "--gaussians 100000 --no-loss --seed 1002" ""; do
    if [ $hw = default ]; then a=""; else a="--host-wait $hw"; fi
    python - $hw $a $cfg <<'PY'
import json, os, subprocess, sys, resource
hw = sys.argv[1]
cmd = [sys.executable, "bench.py", "--no-extra", "--no-cpu-baseline", "--no-other-lists", "--steps", "400"] + sys.argv[2:]
p = subprocess.run(cmd, capture_output=True, text=True)
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
print(f"policy {hw:12s} N={d['config']['n_gaussians']:8d}  ms/step {d['ms_per_step']:.4f}  median {d['ms_per_step_median']:.4f}  "
      f"process CPU {ru.ru_utime + ru.ru_stime:.2f} s (whole run: imports, scene, 400+8 steps)")
PY
  done
done
