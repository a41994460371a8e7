set -x
O=gpurun_out/r04be; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fuzz_regressions.py tests/test_gpu_parity.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log
python tools/experiments/edge4434_probe.py 2>&1 | grep -v amdgpu
