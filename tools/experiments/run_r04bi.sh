set -x
O=gpurun_out/r04bi; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_prevodd.so" timeout 900 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_prevodd.so" timeout 900 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-120
