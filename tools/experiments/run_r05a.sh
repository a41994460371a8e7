set -x
O=gpurun_out/r05a; mkdir -p $O
tools/bin/wrt > $O/wrt.txt 2>&1; cat $O/wrt.txt | tail -8
GSR_AB_LIBS="tools/bin/libgsr_mfma.so" timeout 600 bash tools/ab.sh --steps 20 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-220
GSR_AB_LIBS="tools/bin/libgsr_mfma.so" timeout 600 bash tools/kernel_times.sh --steps 20 --warmup 5 --steady-steps 0 > $O/ktimes.txt 2>&1
grep -E "==|composite_bwd|sort_composite" $O/ktimes.txt
GSR_HIP_LIB=$PWD/tools/bin/libgsr_mfma.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_scale.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
