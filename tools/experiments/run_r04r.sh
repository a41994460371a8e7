set -x
O=gpurun_out/r04r; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_ilp_ssim.so tools/bin/libgsr_ilp_all.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_sched.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_ilp_all.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_sched.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_ilp_all.so" bash tools/ab.sh --steps 40 --warmup 5 --steady-steps 0 --mode rgbd >> $O/ab_sched.txt 2>&1
cat $O/ab_sched.txt
