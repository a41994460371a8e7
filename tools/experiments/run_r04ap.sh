set -x
O=gpurun_out/r04aq; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_touch.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_touch.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt
