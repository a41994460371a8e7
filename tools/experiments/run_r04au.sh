set -x
O=gpurun_out/r04au; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_kept.so" timeout 900 bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_kept.so" timeout 900 bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-110
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log
