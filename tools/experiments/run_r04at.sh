set -x
O=gpurun_out/r04at; mkdir -p $O
# timing-only builds of the aggregating preprocess with one phase removed (results wrong on purpose: only preprocess's own time counts)
#   skip2: no global atomics (positions = counts)   skip3: no second walk at all   skip4: second walk without key stores
GSR_AB_LIBS="tools/bin/libgsr_skip2.so tools/bin/libgsr_skip3.so tools/bin/libgsr_skip4.so" timeout 900 bash tools/ab.sh --steps 40 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-100
