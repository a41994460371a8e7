set -x
O=gpurun_out/r04bd; mkdir -p $O
python tools/experiments/edge4434_probe.py > $O/probe_flat.txt 2>&1
GSR_HIP_LIB=$PWD/tools/bin/libgsr_noflat.so python tools/experiments/edge4434_probe.py > $O/probe_noflat.txt 2>&1
GSR_PREPROCESS_AGG=1 python tools/experiments/edge4434_probe.py > $O/probe_agg.txt 2>&1
grep -v amdgpu.ids $O/probe_flat.txt $O/probe_noflat.txt $O/probe_agg.txt
