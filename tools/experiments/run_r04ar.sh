set -x
O=gpurun_out/r04ar; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_nop1.so tools/bin/libgsr_nop2.so tools/bin/libgsr_nop4.so tools/bin/libgsr_nop8.so tools/bin/libgsr_al64.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt
