set -x
O=gpurun_out/r04s; mkdir -p $O
# in-tree library = max-ilp on composite.hip; the variants add one more scheduler knob each
GSR_AB_LIBS="tools/bin/libgsr_base.so tools/bin/libgsr_ilp_trk.so tools/bin/libgsr_ilp_nounc.so tools/bin/libgsr_ilp_nopost.so tools/bin/libgsr_ilp_relax.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_sched2.txt 2>&1
cat $O/ab_sched2.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
