set -x
O=gpurun_out/r04q; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_maxmem.so tools/bin/libgsr_bias0.so tools/bin/libgsr_bias100.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_sched.txt 2>&1
cat $O/ab_sched.txt
