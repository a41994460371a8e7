set -x
O=gpurun_out/r04aj; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_densify.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python bench.py > $O/bench_line.json 2> $O/bench_line.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04aj/bench_line.json").read().strip().splitlines()[-1])
e=d["extra_configs"]
print("headline", d["ms_per_step"], "morton", e.get("morton_order",{}).get("ms_per_step"), e.get("morton_order",{}).get("stages_ms"), "wall", d.get("bench_wall_s"))
PY
