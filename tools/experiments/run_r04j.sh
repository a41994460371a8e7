set -x
O=gpurun_out/r04j; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --with-optimizer --tail-in-backward >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 >> $O/ab_jac.txt 2>&1
cat $O/ab_jac.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
