#!/bin/bash
# Round-4 one-off GPU-box scripts, one shell function each (they were ~70 files `run_r04<case>.sh`; profiles/README.md and
# profiles/r04/experiments/*.txt cite them as the provenance of their numbers).  On the GPU box, from the repo root:
#   gpurun -- bash tools/experiments/r04_cases.sh <case>
# `--list` prints the cases.  Each body is verbatim what ran; most compare an A/B library built under tools/bin/ (here, on
# the CPU container) with the default build through tools/ab.sh / tools/kernel_times.sh.
cd "$(dirname "$0")/../.."

case_fuzz() {
set -x
O=gpurun_out/r04_fuzz; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py 2500 12 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 1200 python tools/fuzz_parity.py deep 800 0 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 1200 python tools/fuzz_parity.py edge 1500 0 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
timeout 600 python tools/fuzz_parity.py ssim 300 0 > $O/ssim.txt 2>&1; echo "rc=$?" >> $O/ssim.txt
timeout 600 python tools/fuzz_parity.py trainer 40 0 > $O/trainer.txt 2>&1; echo "rc=$?" >> $O/trainer.txt
tail -3 $O/sweep.txt $O/deep.txt $O/edge.txt $O/ssim.txt $O/trainer.txt
}

case_fuzz_agg() {
set -x
# the round's second campaign: the aggregating form of preprocess FORCED (small scenes take the direct form by default)
export GSR_PREPROCESS_AGG=1
O=gpurun_out/r04_fuzz_agg; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1200 3000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 700 python tools/fuzz_parity.py deep 300 1000 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 2000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
tail -3 $O/sweep.txt $O/deep.txt $O/edge.txt
}

case_fuzz_final() {
set -x
# fourth campaign of the round: the FINAL build in its DEFAULT configuration (the fuzz scenes are small: direct form of preprocess,
# now with the flattened walk; grids of odd width take the per-lane walk)
O=gpurun_out/r04_fuzz_final; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1200 7000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 700 python tools/fuzz_parity.py deep 300 2000 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 4000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
for f in sweep deep edge; do tail -n 4 $O/$f.txt; done
}

case_fuzz_flat() {
set -x
# third campaign of the round: the FINAL aggregating preprocess (flattened walks, 16-bit words on large grids) forced
export GSR_PREPROCESS_AGG=1
O=gpurun_out/r04_fuzz_flat; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1200 5000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 700 python tools/fuzz_parity.py deep 300 1500 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 3000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
for f in sweep deep edge; do tail -n 4 $O/$f.txt; done
}

case_fuzz_flat_arb() {
set -x
O=gpurun_out/r04_fuzz_flat; mkdir -p $O
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate sweep 5124 5453 5664 5763  > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 1200 python tools/fuzz_parity.py arbitrate edge 3020 3027 3054 3065 3186 3266 3279 3296 3304 3351 3441  > $O/arb_edge.txt 2>&1
for c in 5124 5453 5664 5763 ; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py 1 $c 2>&1 | tail -2; done > $O/direct_sweep.txt 2>&1
for c in 3020 3027 3054 3065 3186 3266 3279 3296 3304 3351 3441 ; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py edge 1 $c 2>&1 | tail -2; done > $O/direct_edge.txt 2>&1
GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py deep 1 1716 2>&1 | tail -3 > $O/direct_deep.txt
tail -3 $O/arb_sweep.txt $O/arb_edge.txt | cut -c1-200; grep -c "^FAIL" $O/direct_sweep.txt $O/direct_edge.txt $O/direct_deep.txt
}

case_fuzz_last() {
set -x
# last campaign of the round: the final build (flattened walks on grids of odd width too, direct form capped at four waves), default
# configuration, and once more with the aggregating form forced; fresh case ranges
O=gpurun_out/r04_fuzz_last; mkdir -p $O
timeout 700 python tools/fuzz_parity.py 800 9000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 500 python tools/fuzz_parity.py deep 150 2500 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 500 python tools/fuzz_parity.py edge 300 5000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
GSR_PREPROCESS_AGG=1 timeout 700 python tools/fuzz_parity.py 800 9800 > $O/sweep_agg.txt 2>&1; echo "rc=$?" >> $O/sweep_agg.txt
GSR_PREPROCESS_AGG=1 timeout 500 python tools/fuzz_parity.py edge 300 5300 > $O/edge_agg.txt 2>&1; echo "rc=$?" >> $O/edge_agg.txt
for f in sweep deep edge sweep_agg edge_agg; do grep -E "^FAIL|cases passed" $O/$f.txt | awk '{ if ($1=="FAIL") printf "%s:%s ", $3, $NF; else print }'; echo; done
}

case_fuzz_last_arb() {
set -x
O=gpurun_out/r04_fuzz_last; mkdir -p $O
timeout 600 python tools/fuzz_parity.py arbitrate sweep 9647 > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 600 python tools/fuzz_parity.py arbitrate sweep 10071 10339 10494 >> $O/arb_sweep.txt 2>&1
timeout 600 python tools/fuzz_parity.py arbitrate edge 5107 5133 > $O/arb_edge.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate edge 5315 5321 5378 5381 5457 5463 5464 5502 5545 >> $O/arb_edge.txt 2>&1
grep -E "^sweep|^edge|^[0-9]+ / " $O/arb_sweep.txt $O/arb_edge.txt | cut -c1-220
}

case_a() {
set -x
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
tail -3 gpurun_out/r04a/pytest.log
timeout 600 python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err; echo "bench rc=$?"
timeout 300 python tools/experiments/overlap_probe.py > gpurun_out/r04a/overlap.txt 2>&1; cat gpurun_out/r04a/overlap.txt | tail -2
}

case_aa() {
set -x
O=gpurun_out/r04aa; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 direct" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1 GSR_AGG_WAVES=6 python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg6" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1 GSR_AGG_WAVES=4 python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg4" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_ab() {
set -x
O=gpurun_out/r04ab; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
for rep in 1 2; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done; done
for n in 200000 400000 600000 2000000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 30 --warmup 5 --steady-steps 0 --gaussians $n --no-loss 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --mode rgbdn 2>/dev/null | line "rgbdn agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --width 1280 --height 720 --no-loss 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_ac() {
set -x
O=gpurun_out/r04ac; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1 $B 2>/dev/null | line "v2" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v2_norep.so GSR_PREPROCESS_AGG=1 $B 2>/dev/null | line "v2_norep" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b.so GSR_PREPROCESS_AGG=512 $B 2>/dev/null | line "v1b_512" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b.so GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "v1b_direct" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_ad() {
set -x
O=gpurun_out/r04ad; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=512 $B 2>/dev/null | line "replay512" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=256 $B 2>/dev/null | line "replay256" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=1024 $B 2>/dev/null | line "replay1024" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b.so GSR_PREPROCESS_AGG=512 $B 2>/dev/null | line "v1b_512" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
GSR_PREPROCESS_AGG=512 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
}

case_ae() {
set -x
O=gpurun_out/r04ae; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "pytest forms rc=$?"; tail -5 $O/pytest_forms.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
  $B 2>/dev/null | line "default" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_v1b_replay.so GSR_PREPROCESS_AGG=512 $B 2>/dev/null | line "v1b_replay" >> $O/ab.txt 2>&1
done
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for n in 100000 200000 300000 400000 600000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --gaussians $n 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1280 --height 720 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --gaussians 3000000 2>/dev/null | line "1080p_3M agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all.log
}

case_ag() {
set -x
O=gpurun_out/r04ag; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "pytest forms rc=$?"; tail -2 $O/pytest_forms.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | line "scan_under_p2" >> $O/ab.txt 2>&1
  GSR_HIP_LIB=$PWD/tools/bin/libgsr_kept.so $B 2>/dev/null | line "kept" >> $O/ab.txt 2>&1
done
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_ah() {
set -x
O=gpurun_out/r04ah; mkdir -p $O
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate sweep 3127 3285 3648 3661 3808 > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate edge 2003 2157 2223 2356 2466 > $O/arb_edge.txt 2>&1
# the same cases in the direct form: the gradients are bit-identical between the forms, so must be the verdicts
for c in 3127 3285 3648 3661 3808; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py 1 $c 2>&1 | tail -2; done > $O/direct_sweep.txt 2>&1
for c in 2003 2157 2223 2356 2466; do GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py edge 1 $c 2>&1 | tail -2; done > $O/direct_edge.txt 2>&1
GSR_PREPROCESS_AGG=0 timeout 300 python tools/fuzz_parity.py deep 1 1299 2>&1 | tail -3 > $O/direct_deep.txt
tail -30 $O/arb_sweep.txt; tail -30 $O/arb_edge.txt; cat $O/direct_sweep.txt $O/direct_edge.txt $O/direct_deep.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], d['config']['tile_instances'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0"
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B --reference-lists 2>/dev/null | line "reflists agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B --order morton 2>/dev/null | line "morton agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_aj() {
set -x
O=gpurun_out/r04aj; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_densify.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
python bench.py > $O/bench_line.json 2> $O/bench_line.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04aj/bench_line.json").read().strip().splitlines()[-1])
e=d["extra_configs"]
print("headline", d["ms_per_step"], "morton", e.get("morton_order",{}).get("ms_per_step"), e.get("morton_order",{}).get("stages_ms"), "wall", d.get("bench_wall_s"))
PY
}

case_ak() {
set -x
O=gpurun_out/r04ak; mkdir -p $O
for r in 1 2 3 4 5; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 500 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run$r', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], d['steady_state']['ms_per_step'])" >> $O/runs.txt
done
cat $O/runs.txt
}

case_al() {
set -x
O=gpurun_out/r04al; mkdir -p $O
P="import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_max'], (d.get('steady_state') or {}).get('ms_per_step'))"
for r in 1 2 3; do
  python3 bench.py --steps 10 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 300 --mode rgbd 2>/dev/null | python -c "$P" rgbd10 >> $O/runs.txt
  python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 300 --mode rgbd 2>/dev/null | python -c "$P" rgbd20 >> $O/runs.txt
  python3 bench.py --steps 10 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 0 --with-optimizer 2>/dev/null | python -c "$P" opt10 >> $O/runs.txt
  python3 bench.py --steps 10 --warmup 5 --no-extra --no-cpu-baseline --no-other-lists --steady-steps 300 2>/dev/null | python -c "$P" rgb10 >> $O/runs.txt
done
cat $O/runs.txt
}

case_ap() {
set -x
O=gpurun_out/r04aq; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_touch.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_touch.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt
}

case_aq() {
set -x
O=gpurun_out/r04aq; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_touch.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_touch.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt
}

case_ar() {
set -x
O=gpurun_out/r04ar; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_nop1.so tools/bin/libgsr_nop2.so tools/bin/libgsr_nop4.so tools/bin/libgsr_nop8.so tools/bin/libgsr_al64.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt
}

case_as() {
set -x
O=gpurun_out/r04as; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | line "fused" >> $O/ab.txt 2>&1
  GSR_SPLIT_SH=1 $B 2>/dev/null | line "split_sh" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
GSR_SPLIT_SH=1 GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GSR_SPLIT_SH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 20 --steady-steps 0 > $O/prof.log 2>&1
python3 tools/short_kernel_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_split.csv
rm -rf $O/prof
head -6 $O/kernel_stats_split.csv
}

case_at() {
set -x
O=gpurun_out/r04at; mkdir -p $O
# timing-only builds of the aggregating preprocess with one phase removed (results wrong on purpose: only preprocess's own time counts)
#   skip2: no global atomics (positions = counts)   skip3: no second walk at all   skip4: second walk without key stores
GSR_AB_LIBS="tools/bin/libgsr_skip2.so tools/bin/libgsr_skip3.so tools/bin/libgsr_skip4.so" timeout 900 bash tools/ab.sh --steps 40 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-100
}

case_au() {
set -x
O=gpurun_out/r04au; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_kept.so" timeout 900 bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_kept.so" timeout 900 bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-110
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log
}

case_av() {
set -x
O=gpurun_out/r04av; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "forms rc=$?"; tail -3 $O/pytest_forms.log
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | line "flat" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=0 $B 2>/dev/null | line "direct" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_ax() {
set -x
O=gpurun_out/r04ax; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for n in 50000 100000 200000 300000 600000 3000000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --gaussians $n 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1280 --height 720 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --width 2048 --height 1080 --gaussians 1500000 2>/dev/null | line "2048x1080_1.5M agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --order morton 2>/dev/null | line "morton agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --reference-lists 2>/dev/null | line "reflists agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

case_az() {
set -x
O=gpurun_out/r04az; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py -x -q > $O/pytest_forms.log 2>&1; echo "forms rc=$?"; tail -3 $O/pytest_forms.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steady-steps 0 --no-loss"
for rep in 1 2; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --steps 10 --warmup 3 --gaussians 5000000 --width 3840 --height 2160 --seed 1005 2>/dev/null | line "cfg5 agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --steps 30 --warmup 5 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
done; done
cat $O/ab.txt
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_scale.py -x -q > $O/pytest_scale.log 2>&1; echo "scale rc=$?"; grep -E "passed|failed" $O/pytest_scale.log
}

case_b() {
set -x
O=gpurun_out/r04b; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python tools/experiments/overlap_probe.py > $O/overlap.txt 2>&1; tail -2 $O/overlap.txt
}

case_ba() {
set -x
O=gpurun_out/r04ba; mkdir -p $O
export GSR_HIP_LIB=$PWD/tools/bin/libgsr_flatdirect.so
GSR_PREPROCESS_AGG=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest.log 2>&1; echo "parity rc=$?"; grep -E "passed|failed" $O/pytest.log
unset GSR_HIP_LIB
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steady-steps 0 --no-loss"
for rep in 1 2; do for lib in kept flat; do
  if [ $lib = flat ]; then export GSR_HIP_LIB=$PWD/tools/bin/libgsr_flatdirect.so; else unset GSR_HIP_LIB; fi
  GSR_PREPROCESS_AGG=0 $B2 --steps 10 --warmup 3 --gaussians 5000000 --width 3840 --height 2160 --seed 1005 2>/dev/null | line "cfg5 direct_$lib" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=0 $B2 --steps 40 --warmup 5 2>/dev/null | line "cfg3 direct_$lib" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=0 $B2 --steps 40 --warmup 5 --gaussians 50000 2>/dev/null | line "n50k direct_$lib" >> $O/ab.txt 2>&1
done; done
cat $O/ab.txt
}

case_bb() {
set -x
O=gpurun_out/r04bb; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B2="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --no-loss"
for n in 100000 200000 300000 600000 1000000 3000000; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --gaussians $n 2>/dev/null | line "n$n agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1280 --height 720 2>/dev/null | line "720p agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --width 2560 --height 1440 --gaussians 2000000 2>/dev/null | line "1440p_2M agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --order morton 2>/dev/null | line "morton agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m $B2 --reference-lists 2>/dev/null | line "reflists agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all.log
}

case_bd() {
set -x
O=gpurun_out/r04bd; mkdir -p $O
python tools/experiments/edge4434_probe.py > $O/probe_flat.txt 2>&1
GSR_HIP_LIB=$PWD/tools/bin/libgsr_noflat.so python tools/experiments/edge4434_probe.py > $O/probe_noflat.txt 2>&1
GSR_PREPROCESS_AGG=1 python tools/experiments/edge4434_probe.py > $O/probe_agg.txt 2>&1
grep -v amdgpu.ids $O/probe_flat.txt $O/probe_noflat.txt $O/probe_agg.txt
}

case_be() {
set -x
O=gpurun_out/r04be; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fuzz_regressions.py tests/test_gpu_parity.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log
python tools/experiments/edge4434_probe.py 2>&1 | grep -v amdgpu
}

case_bf() {
set -x
O=gpurun_out/r04bf; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do $B 2>/dev/null | line "final" >> $O/ab.txt 2>&1; done
$B --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2" >> $O/ab.txt 2>&1
cat $O/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all.log
}

case_bg() {
set -x
O=gpurun_out/r04bg; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log
GSR_PREPROCESS_AGG=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "agg rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do $B 2>/dev/null | line "cfg3" >> $O/ab.txt 2>&1; done
B2="$B --no-loss"
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m $B2 --width 1896 2>/dev/null | line "odd119 agg$m" >> $O/ab.txt 2>&1
done
$B2 --steps 10 --warmup 3 --gaussians 5000000 --width 3840 --height 2160 --seed 1005 2>/dev/null | line "cfg5" >> $O/ab.txt 2>&1
$B2 --gaussians 100000 --seed 1002 2>/dev/null | line "cfg2" >> $O/ab.txt 2>&1
cat $O/ab.txt
}

case_bi() {
set -x
O=gpurun_out/r04bi; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_prevodd.so" timeout 900 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 > $O/ab.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_prevodd.so" timeout 900 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-120
}

case_bj() {
set -x
O=gpurun_out/r04bj; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_prevodd.so" timeout 900 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-120
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
$B 2>/dev/null | line "cfg3"; $B --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2"
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log
}

case_c() {
set -x
O=gpurun_out/r04c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
for v in default NO_BG0 SSIM_EXACT; do
  case $v in default) E="";; NO_BG0) E="GSR_NO_BG0=1";; SSIM_EXACT) E="GSR_SSIM_EXACT=1";; esac
  for mode in rgb rgbd rgbdn; do
    env $E $B --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v $mode rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('loss_fwd','loss_bwd','composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
}

case_d() {
set -x
O=gpurun_out/r04d; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
run() { # tag env... -- args
  tag=$1; shift; E=""; while [ "$1" != "--" ]; do E="$E $1"; shift; done; shift
  env $E $B "$@" 2>>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$tag', '$*', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
}
for rep in 1 2; do
  run off GSR_BWD_TAIL=0 -- --mode rgb
  run tail1 GSR_BWD_TAIL=1 -- --mode rgb
  run tail2 GSR_BWD_TAIL=2 -- --mode rgb
  run off GSR_BWD_TAIL=0 -- --mode rgbd
  run tail1 GSR_BWD_TAIL=1 -- --mode rgbd
  run tail2 GSR_BWD_TAIL=2 -- --mode rgbd
done
run tail1_s5120 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=5120 -- --mode rgb
run tail1_s5632 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=5632 -- --mode rgb
run tail1_s6656 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=6656 -- --mode rgb
run tail1_s7168 GSR_BWD_TAIL=1 GSR_BWD_SLOTS=7168 -- --mode rgb
run off GSR_BWD_TAIL=0 -- --mode rgbdn
run tail1 GSR_BWD_TAIL=1 -- --mode rgbdn
run off5 GSR_BWD_TAIL=0 -- --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 10
run tail5 GSR_BWD_TAIL=1 -- --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --steps 10
run off2 GSR_BWD_TAIL=0 -- --gaussians 100000 --no-loss --seed 1002
run tail2c GSR_BWD_TAIL=1 -- --gaussians 100000 --no-loss --seed 1002
cat $O/ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
}

case_e() {
set -x
O=gpurun_out/r04e; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/measure_all.sh r04 > $O/measure_all.log 2>&1
tail -3 $O/measure_all.log
}

case_f() {
set -x
O=gpurun_out/r04f; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --with-optimizer --tail-in-backward >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_jac.txt 2>&1
cat $O/ab_jac.txt
}

case_g() {
set -x
O=gpurun_out/r04g; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --with-optimizer --tail-in-backward >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 3 --steady-steps 0 --mode rgbdn >> $O/ab_jac.txt 2>&1
cat $O/ab_jac.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
}

case_h() {
set -x
O=gpurun_out/r04h; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_jac.txt 2>&1
cat $O/ab_jac.txt
timeout 900 python -m pytest tests/test_gpu_fuzz_regressions.py -q -m gpu -s > $O/pytest_fuzz.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fuzz.log
grep -n "sweep\|edge\|needle\|passed\|failed\|^E  " $O/pytest_fuzz.log | cut -c1-400 | head -60
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
}

case_i() {
set -x
O=gpurun_out/r04i; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --with-optimizer --tail-in-backward >> $O/ab_jac.txt 2>&1
cat $O/ab_jac.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
}

case_j() {
set -x
O=gpurun_out/r04j; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --with-optimizer --tail-in-backward >> $O/ab_jac.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_r04_nojac.so" bash tools/ab.sh --steps 30 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 >> $O/ab_jac.txt 2>&1
cat $O/ab_jac.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
}

case_k() {
set -x
O=gpurun_out/r04k; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
bash tools/measure_all.sh r04 > $O/measure_all.log 2>&1
tail -3 $O/measure_all.log
}

case_l() {
set -x
O=gpurun_out/r04l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_scale.py -x -q -m gpu -k "ssim or loss or golden or config3_full_step or rgbd_1m" > $O/pytest_ssim.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ssim.log
tail -4 $O/pytest_ssim.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2; do
for v in tile32 tile16; do
  case $v in tile32) E="GSR_X=1";; tile16) E="GSR_SSIM_TILE16=1";; esac
  for mode in rgb rgbd; do
    env $E $B --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v $mode rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('loss_fwd','loss_bwd','composite_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
}

case_m() {
set -x
O=gpurun_out/r04m; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py arbitrate edge 25 43 51 77 205 257 301 316 358 383 384 395 406 418 420 559 569 576 622 728 803 880 897 931 957 1001 1155 1220 1263 1323 1342 > $O/arbitrate_edge.txt 2>&1; echo "rc=$?" >> $O/arbitrate_edge.txt
grep -c "NOT EXPLAINED" $O/arbitrate_edge.txt; tail -3 $O/arbitrate_edge.txt | cut -c1-300
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
GSR_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 2> $O/bench_1rank_rccl_torchrun.err | grep "^{" > $O/bench_1rank_rccl_torchrun.json
cut -c1-200 $O/bench_1rank_rccl_torchrun.json
}

case_n() {
set -x
O=gpurun_out/r04n; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py arbitrate edge 25 43 51 77 205 257 301 316 358 383 384 395 406 418 420 559 569 576 622 728 803 880 897 931 957 1001 1155 1220 1263 1323 1342 > $O/arbitrate_edge.txt 2>&1; echo "rc=$?" >> $O/arbitrate_edge.txt
grep "NOT EXPLAINED" $O/arbitrate_edge.txt | cut -c1-500; tail -2 $O/arbitrate_edge.txt
timeout 900 python -m pytest tests/test_gpu_fuzz_regressions.py -q -m gpu > $O/pytest_fuzz.log 2>&1; echo "pytest rc=$?" >> $O/pytest_fuzz.log
tail -3 $O/pytest_fuzz.log
}

case_o() {
set -x
O=gpurun_out/r04o; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --mode rgbd"
for rep in 1 2 3; do
for v in base bg0; do
  case $v in base) E="GSR_X=1";; bg0) E="GSR_BG0_RGBD=1";; esac
  env $E $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
done
done
cat $O/ab.txt
}

case_p() {
set -x
O=gpurun_out/r04p; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0"
for rep in 1 2 3; do
for v in base bg0; do
  case $v in base) E="GSR_X=1";; bg0) E="GSR_BG0_RGB=1";; esac
  env $E $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$v rep$rep', d['ms_per_step'], d['ms_per_step_median'], {k:s[k] for k in ('composite_bwd','pergauss_bwd','sort_composite_fwd') if k in s})" >> $O/ab.txt
done
done
cat $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -x -q -m gpu -k "rgbd or depth_and_normal or forward_backward_vs_oracle or long_lists" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
}

case_q() {
set -x
O=gpurun_out/r04q; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_maxmem.so tools/bin/libgsr_bias0.so tools/bin/libgsr_bias100.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_sched.txt 2>&1
cat $O/ab_sched.txt
}

case_r() {
set -x
O=gpurun_out/r04r; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_ilp_ssim.so tools/bin/libgsr_ilp_all.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_sched.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_ilp_all.so" bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 >> $O/ab_sched.txt 2>&1
GSR_AB_LIBS="tools/bin/libgsr_maxilp.so tools/bin/libgsr_ilp_all.so" bash tools/ab.sh --steps 40 --warmup 5 --steady-steps 0 --mode rgbd >> $O/ab_sched.txt 2>&1
cat $O/ab_sched.txt
}

case_s() {
set -x
O=gpurun_out/r04s; mkdir -p $O
# in-tree library = max-ilp on composite.hip; the variants add one more scheduler knob each
GSR_AB_LIBS="tools/bin/libgsr_base.so tools/bin/libgsr_ilp_trk.so tools/bin/libgsr_ilp_nounc.so tools/bin/libgsr_ilp_nopost.so tools/bin/libgsr_ilp_relax.so" bash tools/ab.sh --steps 60 --warmup 5 --steady-steps 0 > $O/ab_sched2.txt 2>&1
cat $O/ab_sched2.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
}

case_u() {
set -x
O=gpurun_out/r04u; mkdir -p $O
python tools/experiments/overlap_probe_fwd.py > $O/probe.txt 2>&1
python tools/experiments/overlap_probe_fwd.py 33554432 >> $O/probe.txt 2>&1
for rep in 1 2; do for m in 0 1 2; do
  GSR_SPLIT_SH=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('split$m', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $O/ab.txt 2>&1
done; done
for m in 0 2; do
  GSR_SPLIT_SH=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('cfg5 split$m', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $O/ab.txt 2>&1
  GSR_SPLIT_SH=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('cfg2 split$m', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))" >> $O/ab.txt 2>&1
done
cat $O/probe.txt $O/ab.txt
GSR_SPLIT_SH=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_split2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_split2.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
GSR_SPLIT_SH=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 20 --steady-steps 0 > $O/prof.log 2>&1
python3 tools/short_kernel_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_split2.csv
find $O/prof -name "*.csv" -size +1M -delete
cat $O/kernel_stats_split2.csv | head -12
}

case_v() {
set -x
O=gpurun_out/r04v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag, env...
  tag=$1; shift
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 30 --steady-steps 0 > $O/prof_$tag.log 2>&1
  python3 tools/short_kernel_stats.py $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/ks_$tag.csv
  rm -rf $O/prof_$tag
  echo "== $tag" >> $O/summary.txt; grep -E "preprocess|bin_inst|sh_color|tile_scan" $O/ks_$tag.csv | cut -d, -f1-4 >> $O/summary.txt
  grep '^{' $O/prof_$tag.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['roofline']['stages_ms'])" >> $O/summary.txt
  for kv in "$@"; do unset ${kv%%=*}; done
}
run base
run emit8 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8
run emit4 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=4
run emit2 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=2
run emit4_sh2 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=4 GSR_SPLIT_SH=2
cat $O/summary.txt
GSR_SPLIT_EMIT=1 GSR_BIN_PEND=4 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_emit4.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_emit4.log
}

case_w() {
set -x
O=gpurun_out/r04w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # tag, env...
  tag=$1; shift
  for kv in "$@"; do export $kv; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o b -- python3 bench.py --in-process --no-cpu-baseline --no-extra --no-other-lists --steps 30 --steady-steps 0 > $O/prof_$tag.log 2>&1
  python3 tools/short_kernel_stats.py $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/ks_$tag.csv
  rm -rf $O/prof_$tag
  echo "== $tag" >> $O/summary.txt; grep -E "preprocess|bin_inst|sh_color|tile_scan|sort_comp" $O/ks_$tag.csv | cut -d, -f1-4 >> $O/summary.txt
  for kv in "$@"; do unset ${kv%%=*}; done
}
run emit8 GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8
run emit8_repl GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8 GSR_BIN_REPL=8192
run emit8_again GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8
run emit8_repl_again GSR_SPLIT_EMIT=1 GSR_BIN_PEND=8 GSR_BIN_REPL=8192
cat $O/summary.txt
hipcc --offload-arch=gfx950 -O3 tools/atomic_rates.hip -o /tmp/atomic_rates && /tmp/atomic_rates > $O/atomic_rates.txt 2>&1; cat $O/atomic_rates.txt
}

case_y() {
set -x
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_agg.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
for rep in 1 2; do for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 1; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 2>/dev/null | line "cfg5 agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2 agg$m" >> $O/ab.txt 2>&1
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 40 --warmup 5 --steady-steps 0 --mode rgbd 2>/dev/null | line "rgbd agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
GSR_PREPROCESS_AGG=1 timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all_agg.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_all_agg.log
}

case_z() {
set -x
O=gpurun_out/r04z2; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
for rep in 1 2; do for m in 0 256 512 1024; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 2>/dev/null | line "cfg3 agg$m" >> $O/ab.txt 2>&1
done; done
for m in 0 512; do
  GSR_PREPROCESS_AGG=$m python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 60 --warmup 5 --steady-steps 0 --gaussians 100000 --no-loss --seed 1002 2>/dev/null | line "cfg2 agg$m" >> $O/ab.txt 2>&1
done
cat $O/ab.txt
}

if [ "$1" = "--list" ] || [ -z "$1" ]; then declare -F | sed -n "s/^declare -f case_//p"; exit 0; fi
if ! declare -F "case_$1" > /dev/null; then echo "unknown case $1 (try --list)" >&2; exit 2; fi
"case_$1"
