#!/bin/bash
# Round-5 GPU-box scripts, one shell function per case (what profiles/r05/** cite as the provenance of their numbers).
#   gpurun -- bash tools/experiments/r05_cases.sh <case>        (--list prints the cases)
cd "$(dirname "$0")/../.."

# a: composite_bwd with the row stage of its reduction on the matrix pipe (tools/bin/libgsr_mfma.so = composite.hip built with
#    -DGSR_BWD_MFMA) against the default build: unit check of the network, step A/B, rocprofv3 kernel averages, parity suite
case_a() {
set -x
O=gpurun_out/r05a; mkdir -p $O
tools/bin/wrt > $O/wrt.txt 2>&1; cat $O/wrt.txt | tail -8
GSR_AB_LIBS="tools/bin/libgsr_mfma.so" timeout 600 bash tools/ab.sh --steps 20 --warmup 5 --steady-steps 0 > $O/ab.txt 2>&1
grep -E "^(default|tools)" $O/ab.txt | cut -c1-220
GSR_AB_LIBS="tools/bin/libgsr_mfma.so" timeout 600 bash tools/kernel_times.sh --steps 20 --warmup 5 --steady-steps 0 > $O/ktimes.txt 2>&1
grep -E "==|composite_bwd|sort_composite" $O/ktimes.txt
GSR_HIP_LIB=$PWD/tools/bin/libgsr_mfma.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_scale.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
}

# b: MFMA-beside-VALU micro-benchmark; the new GPU tests of the round; the driver line with extra_configs.scenes
case_b() {
set -x
O=gpurun_out/r05b; mkdir -p $O
tools/bin/mfma_valu_overlap > $O/mfma_valu_overlap.txt 2>&1; cat $O/mfma_valu_overlap.txt
timeout 1500 python -m pytest tests/test_gpu_handle_switches.py tests/test_gpu_forward_only.py tests/test_gpu_scenes.py tests/test_gpu_densify.py tests/test_gpu_preprocess_forms.py -x -q --durations=15 > $O/pytest.log 2>&1; echo "rc=$?"; tail -30 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
p=json.loads(open('gpurun_out/r05b/bench.json').read().strip().splitlines()[-1])
print('headline', p['ms_per_step'], p['roofline']['stages_ms'], 'wall', p.get('bench_wall_s'))
for k,v in (p.get('extra_configs',{}).get('scenes',{}) or {}).items():
    if 'ms_per_step' not in v: print(k, v); continue
    c=v.get('vs_config3_cost',{})
    print(k, v['ms_per_step'], 'D', v['tile_instances'], 'V', v['visible'], 'max', v['max_tile_instances'], 'compact', v['compact_binning'], v['preprocess_form'], 'wall', v.get('wall_s'))
    print('   pred', c.get('predicted_ms_per_step'), 'ratio', c.get('ratio'), 'over', c.get('stages_over_bar'))
    for st,x in c.get('stages',{}).items(): print('     ', st, x)
PY
}

# c: per-kernel times (rocprofv3 --kernel-trace --stats) of the non-uniform scenes of extra_configs.scenes
case_c() {
set -x
O=gpurun_out/r05c; mkdir -p $O
for sc in "hot --skew hot:32000 --no-loss" "dense4k --gaussians 5000000 --width 3840 --height 2160 --seed 1005 --skew dense:0.01:50 --no-loss" "trained1m --scene trained --seed 1010 --mode rgbd" "trained3m --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440"; do
  set -- $sc; tag=$1; shift
  timeout 600 bash tools/kernel_times.sh --steps 10 --warmup 3 --steady-steps 0 "$@" > $O/ktimes_$tag.txt 2>&1
  echo "== $tag"; grep -E "avg" $O/ktimes_$tag.txt
  cp gpurun_out/ktimes_default/st_kernel_stats.csv $O/kernel_stats_$tag.csv
done
}

# d: pergauss_bwd with ∇scales / ∇rotations in float64 (default build) against the fp32 chain (tools/bin/libgsr_pgb32.so =
#    pergauss.hip built with -DGSR_PGB_FP32_CHAIN): step A/B at configs 3 and 5, then the whole -m gpu suite on the default build
case_d() {
set -x
O=gpurun_out/r05d; mkdir -p $O
GSR_AB_LIBS="tools/bin/libgsr_pgb32.so" timeout 600 bash tools/ab.sh --steps 20 --warmup 5 --steady-steps 0 > $O/ab_cfg3.txt 2>&1
grep -E "^(default|tools)" $O/ab_cfg3.txt | cut -c1-220
GSR_AB_LIBS="tools/bin/libgsr_pgb32.so" timeout 600 bash tools/ab.sh --steps 10 --warmup 3 --steady-steps 0 --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 > $O/ab_cfg5.txt 2>&1
grep -E "^(default|tools)" $O/ab_cfg5.txt | cut -c1-220
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "rc=$?"; tail -25 $O/pytest.log
grep -E "fp32 restatement|HIP-f64|needle" $O/pytest.log | head -20
}

# e: the banded aggregating form + the SCATTER pass of the compact mode: parity (forms test, parity, scenes), then timing —
#    config 5 default (direct) vs forced aggregating (banded), dense 4K, the hot tile and the 3 M trained-like scene (before: the
#    same scenes in cases b / c of this file, round-4 binning)
case_e() {
set -x
O=gpurun_out/r05e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_preprocess_forms.py tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_handle_switches.py -x -q --durations=8 > $O/pytest.log 2>&1; echo "rc=$?"; tail -16 $O/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
for rep in 1 2; do
$B $C5 2>/dev/null | line "cfg5 default";
GSR_PREPROCESS_AGG=1 $B $C5 2>/dev/null | line "cfg5 agg-banded";
done
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k default"
GSR_PREPROCESS_AGG=1 $B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k agg-banded"
$B --skew hot:32000 --no-loss 2>/dev/null | line "hot32k default"
$B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 2>/dev/null | line "trained3m default"
$B 2>/dev/null | line "cfg3 default"; $B 2>/dev/null | line "cfg3 default"
}

# f: long tiles — forward strip kernel with the next chunk's mask plane prefetched, listed backward with 256-splat batches and
#    the next batch in flight; the skew hint for banded binning at 4K; parity of the long-list paths; kernel times of the hot scene
case_f() {
set -x
O=gpurun_out/r05f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_preprocess_forms.py tests/test_gpu_forward_only.py tests/test_gpu_fuzz_regressions.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
$B --skew hot:32000 --no-loss 2>/dev/null | line "hot32k"
$B --skew hot:8000 --no-loss 2>/dev/null | line "hot8k"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k"
$B $C5 2>/dev/null | line "cfg5"
$B --scene trained --seed 1010 --mode rgbd 2>/dev/null | line "trained1m"
$B 2>/dev/null | line "cfg3"
timeout 600 bash tools/kernel_times.sh --steps 10 --warmup 3 --steady-steps 0 --skew hot:32000 --no-loss > $O/ktimes_hot.txt 2>&1
cp gpurun_out/ktimes_default/st_kernel_stats.csv $O/kernel_stats_hot.csv
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r05f/kernel_stats_hot.csv',newline='')))
for r in rows[:12]:
    m=re.search(r'(\w+_kernel)',r['Name']); print('  %-34s calls %4s avg %9.1f us'%((m.group(1) if m else r['Name'][:34]), r['Calls'], float(r['AverageNs'])/1e3))
PY
}

# g: long tiles' backward split along the list (composite_bwd_long_kernel: first 8 segment waves in ONE workgroup per tile —
#    1.77 -> 1.24 ms, bound by the four SIMDs of its CU —, then one single-wave workgroup per (tile, segment), 32 segments, two
#    launches) against round 4's
#    four strips per tile (GSR_BWD_LONG=0): parity of every long-list test, then hot-tile scenes both ways
case_g() {
set -x
O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_scale.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
for sk in hot:32000 hot:8000 hot:128000; do
  $B --skew $sk --no-loss 2>/dev/null | line "$sk by-list-segments"
  GSR_BWD_LONG=0 $B --skew $sk --no-loss 2>/dev/null | line "$sk by-strips"
done
$B --scene trained --seed 1010 --mode rgbd 2>/dev/null | line "trained1m by-list-segments"
GSR_BWD_LONG=0 $B --scene trained --seed 1010 --mode rgbd 2>/dev/null | line "trained1m by-strips"
$B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 2>/dev/null | line "trained3m by-list-segments"
GSR_BWD_LONG=0 $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 2>/dev/null | line "trained3m by-strips"
$B 2>/dev/null | line "cfg3"
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k split<=256 tiles"
GSR_BWD_SPLIT_TILES=4096 $B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k split<=4096 tiles"
GSR_BWD_SPLIT_TILES=4096 $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 2>/dev/null | line "trained3m split<=4096"
GSR_BWD_SPLIT_TILES=4096 $B --scene trained --seed 1010 --mode rgbd 2>/dev/null | line "trained1m split<=4096"
}

# h: lists beyond 8192 keys sorted by many workgroups (plan -> chunk sorts -> merge passes -> emit as separate launches) instead of
#    one 1024-thread workgroup per tile: parity (every test with a long list), then the hot-tile scenes and dense 4K
case_h() {
set -x
O=gpurun_out/r05h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_preprocess_forms.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
for sk in hot:32000 hot:8000 hot:128000; do $B --skew $sk --no-loss 2>/dev/null | line "$sk"; done
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k"
$B 2>/dev/null | line "cfg3"
timeout 600 bash tools/kernel_times.sh --steps 10 --warmup 3 --steady-steps 0 --skew hot:32000 --no-loss > $O/ktimes_hot.txt 2>&1
cp gpurun_out/ktimes_default/st_kernel_stats.csv $O/kernel_stats_hot.csv
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r05h/kernel_stats_hot.csv',newline='')))
for r in rows[:18]:
    m=re.search(r'(\w+_kernel)',r['Name']); print('  %-34s calls %4s avg %9.1f us'%((m.group(1) if m else r['Name'][:34]), r['Calls'], float(r['AverageNs'])/1e3))
PY
}

# fuzz: the round's campaign on the build with the banded / SCATTER binning, the float64 scales-rotations chain, the list-split long
#       backward and the multi-workgroup long sort: fresh case ranges; once more with the aggregating form forced
case_fuzz() {
set -x
O=gpurun_out/r05_fuzz; mkdir -p $O
timeout 900 python tools/fuzz_parity.py 1000 12000 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 900 python tools/fuzz_parity.py deep 400 3000 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 700 python tools/fuzz_parity.py edge 500 6000 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
GSR_PREPROCESS_AGG=1 timeout 700 python tools/fuzz_parity.py 600 13000 > $O/sweep_agg.txt 2>&1; echo "rc=$?" >> $O/sweep_agg.txt
GSR_PREPROCESS_AGG=1 timeout 700 python tools/fuzz_parity.py deep 200 3400 > $O/deep_agg.txt 2>&1; echo "rc=$?" >> $O/deep_agg.txt
timeout 300 python tools/fuzz_parity.py trainer 40 100 > $O/trainer.txt 2>&1; echo "rc=$?" >> $O/trainer.txt
for f in sweep deep edge sweep_agg deep_agg trainer; do echo "== $f"; grep -E "^FAIL|cases passed|^rc=" $O/$f.txt | awk '{ if ($1=="FAIL") printf "%s:%s ", $3, $NF; else print }'; echo; done
}

# i: float64 arbitration of the campaign's failing cases; the tier launches beside the fused forward (GSR_NO_TIER_OVERLAP=1 = before);
#    the whole -m gpu suite on the build
case_i() {
set -x
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python tools/fuzz_parity.py arbitrate sweep 12490 12526 > $O/arb_sweep.txt 2>&1
GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py arbitrate sweep 13123 13257 13517 13598 > $O/arb_sweep_agg.txt 2>&1
timeout 900 python tools/fuzz_parity.py arbitrate edge 6389 6411 6426 6432 > $O/arb_edge.txt 2>&1
grep -hE "^(sweep|edge) [0-9]+|Assertion|Error" $O/arb_*.txt | cut -c1-700
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 --steady-steps 0"
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
for rep in 1 2; do
$B --scene trained --seed 1010 --mode rgbd 2>/dev/null | line "trained1m beside"
GSR_NO_TIER_OVERLAP=1 $B --scene trained --seed 1010 --mode rgbd 2>/dev/null | line "trained1m behind"
done
$B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 2>/dev/null | line "trained3m beside"
GSR_NO_TIER_OVERLAP=1 $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 2>/dev/null | line "trained3m behind"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k beside"
GSR_NO_TIER_OVERLAP=1 $B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k behind"
$B 2>/dev/null | line "cfg3"
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -4 $O/pytest.log
}

# j: the handle's second stream at the highest priority: tier launches beside the fused forward (GSR_NO_TIER_OVERLAP=1 = behind it),
#    GSR_AUX_STREAM_DEFAULT_PRIORITY=1 = the stream as in rounds 2-4
case_j() {
set -x
O=gpurun_out/r05j; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 --steady-steps 0"
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
T1="--scene trained --seed 1010 --mode rgbd"; T3="--scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440"
for rep in 1 2; do
$B $T1 2>/dev/null | line "trained1m beside, high-priority stream"
GSR_AUX_STREAM_DEFAULT_PRIORITY=1 $B $T1 2>/dev/null | line "trained1m beside, default priority"
GSR_NO_TIER_OVERLAP=1 $B $T1 2>/dev/null | line "trained1m behind"
done
$B $T3 2>/dev/null | line "trained3m beside, high-priority stream"
GSR_NO_TIER_OVERLAP=1 $B $T3 2>/dev/null | line "trained3m behind"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k beside, high-priority stream"
GSR_NO_TIER_OVERLAP=1 $B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k behind"
$B --skew hot:32000 --no-loss 2>/dev/null | line "hot32k high-priority stream"
GSR_AUX_STREAM_DEFAULT_PRIORITY=1 $B --skew hot:32000 --no-loss 2>/dev/null | line "hot32k default priority"
$B 2>/dev/null | line "cfg3"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_forward_only.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
}

# k: the forward of lists beyond 8192 instances split along the list (composite_fwd_long_kernel, two launches on the second stream beside
#    the strip launch; GSR_FWD_LONG=0 = before): parity of every long-list test, forward-only renders, then the hot-tile scenes
case_k() {
set -x
O=gpurun_out/r05k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/fuzz_parity.py deep 150 3600 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed" $O/deep.txt | cut -c1-200
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 --steady-steps 0"
for sk in hot:32000 hot:128000 hot:8000; do
  $B --skew $sk --no-loss 2>/dev/null | line "$sk fwd by list segments"
  GSR_FWD_LONG=0 $B --skew $sk --no-loss 2>/dev/null | line "$sk fwd by quadrants"
done
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k"
$B 2>/dev/null | line "cfg3"
}

# l: the forward of lists beyond 8192 instances by sixteen 4x4-block waves per tile on the second stream (GSR_FWD_BLOCKS=0 = before):
#    parity incl. the bit-identity of the two list modes (deep fuzz), then the hot-tile scenes
case_l() {
set -x
O=gpurun_out/r05l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/fuzz_parity.py deep 200 3800 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed" $O/deep.txt | cut -c1-200
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stages_ms']
print('$1', d['ms_per_step'], 'D', d['config']['tile_instances'], d['config']['binning']['mode'][:7], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))"; }
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 --steady-steps 0"
for sk in hot:32000 hot:128000 hot:8000; do
  $B --skew $sk --no-loss 2>/dev/null | line "$sk fwd long tiles by 4x4 blocks"
  GSR_FWD_BLOCKS=0 $B --skew $sk --no-loss 2>/dev/null | line "$sk fwd by quadrants"
done
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
$B $C5 --skew dense:0.01:50 2>/dev/null | line "dense4k"
$B 2>/dev/null | line "cfg3"
}

# m: trained-like scenes with LONGER lists (in-plane splat size 8 / 12 px instead of 4): how much of a scene like a real capture is
#    beyond the fused forward's 1024-instance cut, and what the tier path costs there
case_m() {
set -x
O=gpurun_out/r05m; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
for sg in 4 8 12; do
  $B --scene trained --seed 1010 --mode rgbd --sigma-px $sg 2>/dev/null > $O/t1m_$sg.json
  python - $O/t1m_$sg.json $sg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print('trained1m sigma', sys.argv[2], 'ms', d['ms_per_step'], 'D', c['tile_instances'], 'V', c['visible'], 'longest', c['binning']['longest_tile_list'], c['binning']['mode'][:8], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
done
}

# n: the (1024, 8192] tier sorts by register runs + LDS merges (tile_sort_runs_kernel; GSR_SORT_TIERS_NETWORK=1 = round 2's LDS network):
#    parity (every test with mid-length lists + 200 deep fuzz scenes), then trained-like scenes with longer lists and dense 4K
case_n() {
set -x
O=gpurun_out/r05n; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py tests/test_gpu_scale.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/fuzz_parity.py deep 200 4000 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed" $O/deep.txt | cut -c1-200
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'D', c['tile_instances'], 'longest', c['binning']['longest_tile_list'], c['binning']['mode'][:8], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
for sg in 4 8 12; do
  run "trained1m sigma $sg runs+merges" $B --scene trained --seed 1010 --mode rgbd --sigma-px $sg
  GSR_SORT_TIERS_NETWORK=1 run "trained1m sigma $sg LDS network" $B --scene trained --seed 1010 --mode rgbd --sigma-px $sg
done
run "trained3m runs+merges" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
GSR_SORT_TIERS_NETWORK=1 run "trained3m LDS network" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
run "dense4k runs+merges" $B $C5 --skew dense:0.01:50
GSR_SORT_TIERS_NETWORK=1 run "dense4k LDS network" $B $C5 --skew dense:0.01:50
run "hot8k runs+merges" $B --skew hot:8000 --no-loss
GSR_SORT_TIERS_NETWORK=1 run "hot8k LDS network" $B --skew hot:8000 --no-loss
run "cfg3" $B
}

# o: the (1024, 4096] tier of a fused-path view sorted AND composited by one launch (sort_composite_fwd_mid_kernel; GSR_NO_MID_FUSED=1 =
#    tier sort + strip kernel): parity (+ 200 deep fuzz scenes), trained-like scenes with in-plane splat sizes 4 / 8 / 12 px, 3 M, dense 4K
case_o() {
set -x
O=gpurun_out/r05o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_fuzz_regressions.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py tests/test_gpu_scale.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
timeout 600 python tools/fuzz_parity.py deep 200 4200 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed" $O/deep.txt | cut -c1-200
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'D', c['tile_instances'], 'longest', c['binning']['longest_tile_list'], c['binning']['mode'][:8], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
for sg in 4 8 12; do
  run "trained1m sigma $sg mid fused" $B --scene trained --seed 1010 --mode rgbd --sigma-px $sg
  GSR_NO_MID_FUSED=1 run "trained1m sigma $sg tier sort + strip" $B --scene trained --seed 1010 --mode rgbd --sigma-px $sg
done
run "trained3m mid fused" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
GSR_NO_MID_FUSED=1 run "trained3m tier sort + strip" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
run "dense4k mid fused" $B $C5 --skew dense:0.01:50
GSR_NO_MID_FUSED=1 run "dense4k tier sort + strip" $B $C5 --skew dense:0.01:50
run "cfg3" $B
}

# p: bins + overflow tiles (a few deep tiles no longer send the whole view to the compact mode): parity, then the skewed scenes with
#    the default budget (bins + overflow scatter), a budget of 1 byte (compact mode, as before) and, hot tile only, a budget that
#    holds bins for the deep tile
case_p() {
set -x
O=gpurun_out/r05p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py tests/test_gpu_handle_switches.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'D', c['tile_instances'], 'longest', c['binning']['longest_tile_list'], c['binning']['mode'][:12], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
for bb in 0 1; do
  run "hot32k budget $bb" $B --skew hot:32000 --seed 1003 --bins-budget $bb
  run "hot128k budget $bb" $B --skew hot:128000 --seed 1003 --bins-budget $bb
  run "dense4k budget $bb" $B --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --skew dense:0.01:50 --bins-budget $bb
  run "trained3m budget $bb" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440 --bins-budget $bb
done
run "hot32k budget 4G" $B --skew hot:32000 --seed 1003 --bins-budget 4000000000
run "cfg3" $B
timeout 900 python tools/fuzz_parity.py deep 200 4300 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed" $O/deep.txt | cut -c1-200
}

# q: the long-list chain BESIDE the fused launch on reserved CUs (CU-masked streams) when the tier tiles are few: hot tile scenes
#    with 0 (off) / 8 / 16 / 32 reserved CUs, and the throughput-bound scenes at two limits
case_q() {
set -x
O=gpurun_out/r05q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_forward_only.py tests/test_gpu_preprocess_forms.py -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -5 $O/pytest.log
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'D', c['tile_instances'], 'longest', c['binning']['longest_tile_list'], c['binning']['mode'][:24], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
for mx in 64 100000; do
  export GSR_TIERS_BESIDE_MAX=$mx
  run "hot32k beside_max=$mx" $B --skew hot:32000 --seed 1003
  run "hot32k no loss beside_max=$mx" $B --skew hot:32000 --seed 1003 --no-loss
  run "hot128k beside_max=$mx" $B --skew hot:128000 --seed 1003
  run "hot6k beside_max=$mx" $B --skew hot:6000 --seed 1003
  run "trained1m max=$mx" $B --scene trained --seed 1010 --mode rgbd
  run "trained1m sigma 12 max=$mx" $B --scene trained --seed 1010 --mode rgbd --sigma-px 12
  run "trained3m max=$mx" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
  run "dense4k max=$mx" $B --gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005 --skew dense:0.01:50
  run "dense1080 max=$mx" $B --skew dense:0.02:20 --seed 1003
done
GSR_WALK_PRIO=0 run "hot32k walk at normal priority" $B --skew hot:32000 --seed 1003
GSR_WALK_PRIO=0 run "trained3m walk at normal priority" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
GSR_SORT_TIERS_NETWORK=1 run "hot32k network sorts" $B --skew hot:32000 --seed 1003
unset GSR_TIERS_BESIDE_MAX
run "hot32k compact" $B --skew hot:32000 --seed 1003 --bins-budget 1
run "cfg3" $B
timeout 900 python tools/fuzz_parity.py deep 150 4500 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed" $O/deep.txt | cut -c1-200
}

# r: kernel timeline (rocprofv3 --kernel-trace) of one hot-tile step with the chain on reserved CUs / without
case_r() {
O=gpurun_out/r05r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd - > /dev/null
for r in 64 0; do
  export GSR_TIERS_BESIDE_MAX=$r
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$r -- python3 bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --no-scenes --steps 4 --warmup 3 --steady-steps 0 --skew hot:32000 --seed 1003 > $O/bench_$r.log 2>&1
  tail -1 $O/bench_$r.log | cut -c1-200
  python3 tools/experiments/timeline.py $O/trace_$r > $O/timeline_$r.txt 2>&1; cat $O/timeline_$r.txt | head -60
  rm -rf $O/trace_$r
done
}

# s: full GPU suite + the fuzz families on the build with overflow tiles / the held fused launch, incl. deep scenes under a bins
#    budget of 1024 and 1536 keys per tile (second view: bins + overflow lists) in both binning forms
case_s() {
O=gpurun_out/r05s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
timeout 900 python tools/fuzz_parity.py deep 300 5000 > $O/deep.txt 2>&1; grep -E "^FAIL|cases passed|binning mode" $O/deep.txt | cut -c1-220
GSR_FUZZ_BINS_KEYS=1024 timeout 900 python tools/fuzz_parity.py deep 300 5300 > $O/deep_1024.txt 2>&1; grep -E "^FAIL|cases passed|binning mode" $O/deep_1024.txt | cut -c1-220
GSR_FUZZ_BINS_KEYS=1536 GSR_PREPROCESS_AGG=1 timeout 900 python tools/fuzz_parity.py deep 300 5600 > $O/deep_1536_agg.txt 2>&1; grep -E "^FAIL|cases passed|binning mode" $O/deep_1536_agg.txt | cut -c1-220
GSR_FUZZ_BINS_KEYS=1024 timeout 600 python tools/fuzz_parity.py edge 300 7000 > $O/edge_1024.txt 2>&1; grep -E "^FAIL|cases passed|binning mode" $O/edge_1024.txt | cut -c1-220
timeout 600 python tools/fuzz_parity.py 400 14000 > $O/sweep.txt 2>&1; grep -E "^FAIL|cases passed" $O/sweep.txt | cut -c1-220
}

# t: what spatial order of the Gaussians buys the binning at 4K (config 5) and at config 3, per binning form
case_t() {
O=gpurun_out/r05t; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'D', c['tile_instances'], 'longest', c['binning']['longest_tile_list'], c['binning']['mode'][:24], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
for ord in random morton; do
  run "cfg5 $ord default form" $B $C5 --order $ord
  GSR_PREPROCESS_AGG=1 run "cfg5 $ord aggregating (banded)" $B $C5 --order $ord
  GSR_PREPROCESS_AGG=0 run "cfg3 $ord direct" $B --order $ord
  GSR_PREPROCESS_AGG=1 run "cfg3 $ord aggregating" $B --order $ord
done
}

# u: the default form choice on banded grids measured per handle (FormTuner): config 5 in random / Morton order, dense 4K, with the
#    tuner and without (GSR_FORM_TUNER=0: the previous view's skew hint alone)
case_u() {
O=gpurun_out/r05u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_preprocess_forms.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 10 --warmup 8 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'form', c['binning'].get('preprocess_form'), 'D', c['tile_instances'], ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
C5="--gaussians 5000000 --width 3840 --height 2160 --no-loss --seed 1005"
for tn in 1 0; do
  export GSR_FORM_TUNER=$tn
  run "cfg5 random tuner=$tn" $B $C5
  run "cfg5 morton tuner=$tn" $B $C5 --order morton
  run "dense4k tuner=$tn" $B $C5 --skew dense:0.01:50
  run "cfg3 tuner=$tn" $B
done
}

# v: register-blocked SSIM kernels (two outputs per thread and pass, 8-byte LDS reads) against the one-output-per-thread ones
#    (measured slower and removed: DESIGN.md §4.2; the kernels are in git history only — GSR_SSIM_TILING no longer exists)
case_v() {
O=gpurun_out/r05v; mkdir -p $O
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_parity.py tests/test_gpu_trainer.py tests/test_gpu_handle_switches.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], ' '.join(f'{k}={v:.4f}' for k,v in s.items()))
PY
}
for v in 0 1 0 1; do
  GSR_SSIM_TILING=$v run "cfg3 tiling=$v" $B
  GSR_SSIM_TILING=$v GSR_SSIM_EXACT=1 run "cfg3 exact tiling=$v" $B
  GSR_SSIM_TILING=$v run "rgbd tiling=$v" $B --mode rgbd
done
}

# w: a held fused launch sends a first slice of its slots out before the host wait (GSR_HELD_EARLY_PERMILLE of the grid)
#    (measured slower — the first slots are the longest lists, and the tier sorts wait for them — and removed: the knob no longer exists)
case_w() {
O=gpurun_out/r05w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_forward_only.py -x -q 2>&1 | tail -2
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --steps 20 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' '.join(f'{k}={v:.3f}' for k,v in s.items()))
PY
}
for pm in 0 60 30 120 0 60; do
  export GSR_HELD_EARLY_PERMILLE=$pm
  run "trained1m early=$pm" $B --scene trained --seed 1010 --mode rgbd
  run "trained3m early=$pm" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
  run "hot32k early=$pm" $B --skew hot:32000 --seed 1003
  run "hot6k early=$pm" $B --skew hot:6000 --seed 1003
done
}

# x: last fuzz campaign of the round on the final build (every family, both binning forms, bins budgets)
case_x() {
O=gpurun_out/r05x; mkdir -p $O
G='^FAIL|cases passed|binning mode'
timeout 900 python tools/fuzz_parity.py 800 15000 > $O/sweep.txt 2>&1; grep -E "$G" $O/sweep.txt | cut -c1-220
timeout 900 python tools/fuzz_parity.py deep 400 6000 > $O/deep.txt 2>&1; grep -E "$G" $O/deep.txt | cut -c1-220
timeout 700 python tools/fuzz_parity.py edge 500 8000 > $O/edge.txt 2>&1; grep -E "$G" $O/edge.txt | cut -c1-220
GSR_PREPROCESS_AGG=1 timeout 700 python tools/fuzz_parity.py 500 16000 > $O/sweep_agg.txt 2>&1; grep -E "$G" $O/sweep_agg.txt | cut -c1-220
GSR_PREPROCESS_AGG=1 GSR_FUZZ_BINS_KEYS=2048 timeout 700 python tools/fuzz_parity.py deep 250 6400 > $O/deep_agg_2048.txt 2>&1; grep -E "$G" $O/deep_agg_2048.txt | cut -c1-220
GSR_TIERS_BESIDE_MAX=0 timeout 600 python tools/fuzz_parity.py deep 200 6700 > $O/deep_not_held.txt 2>&1; grep -E "$G" $O/deep_not_held.txt | cut -c1-220
timeout 300 python tools/fuzz_parity.py trainer 40 200 > $O/trainer.txt 2>&1; grep -E "$G" $O/trainer.txt | cut -c1-220
timeout 300 python tools/fuzz_parity.py ssim 300 1000 > $O/ssim.txt 2>&1; grep -E "$G" $O/ssim.txt | cut -c1-220
}

# y: backward with a colour-only pixel cotangent (what the photometric loss head produces: channels >= 3 are zeros) in :rgbd /
#    :rgbdn — the :rgb arithmetic on the mode's stream
case_y() {
O=gpurun_out/r05y; mkdir -p $O
B="python bench.py --in-process --no-extra --no-cpu-baseline --no-other-lists --no-scenes --steps 20 --warmup 3 --steady-steps 0"
run() { tag=$1; shift; "$@" 2>/dev/null > $O/tmp.json; python - $O/tmp.json "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']; s=d['roofline']['stages_ms']
print(sys.argv[2], 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'), ' '.join(f'{k}={v:.4f}' for k,v in s.items()))
PY
}
for v in 0 1 0 1; do
  GSR_BWD_COLOR_ONLY=$v run "rgbd color_only=$v" $B --mode rgbd
  GSR_BWD_COLOR_ONLY=$v run "rgbdn color_only=$v" $B --mode rgbdn
  GSR_BWD_COLOR_ONLY=$v run "trained3m rgbd color_only=$v" $B --scene trained --seed 1011 --mode rgbd --gaussians 3000000 --width 2560 --height 1440
done
run "rgb" $B
}

if [ "$1" = "--list" ] || [ -z "$1" ]; then declare -F | sed -n "s/^declare -f case_//p"; exit 0; fi
if ! declare -F "case_$1" > /dev/null; then echo "unknown case $1 (try --list)" >&2; exit 2; fi
"case_$1"
