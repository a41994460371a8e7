#!/bin/bash
# Round 6, on the GPU box: the experiment cases whose outputs are kept under profiles/r06/ (each case = one block below).
#   tools/experiments/r06_cases.sh <case>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
O=gpurun_out/r06; mkdir -p $O
case "$1" in
needles)
    # (a) accuracy: HIP vs float64 on the needle scenes, default build (needle instances refined), every instance refined, none
    for v in default needles2 needles0; do
        echo "== $v"
        if [ $v = default ]; then unset GSR_HIP_LIB; else export GSR_HIP_LIB=$PWD/tools/bin/libgsr_$v.so; fi
        timeout 900 python tools/experiments/needle_means.py edge 8498 8112 5315 5457 8421 8437 8283 2>&1 | grep "^edge" | sed "s/vshs.*vscales/... vscales/" | cut -c1-330
    done > $O/needle_means.txt 2>&1
    unset GSR_HIP_LIB
    cat $O/needle_means.txt
    # (b) cost: config 3 (no needles), the trained-like 1 M scene (flat splats: many needles), :rgbd
    GSR_AB_LIBS="tools/bin/libgsr_needles0.so tools/bin/libgsr_needles2.so" tools/ab.sh > $O/needle_ab_cfg3.txt 2>&1
    GSR_AB_LIBS="tools/bin/libgsr_needles0.so tools/bin/libgsr_needles2.so" tools/ab.sh --scene trained --seed 1010 --mode rgbd > $O/needle_ab_trained.txt 2>&1
    cat $O/needle_ab_cfg3.txt $O/needle_ab_trained.txt
    ;;
fuzz)
    # the round's fuzz campaign on the final build: every family, both binning forms, a small bins budget, the fused launch never
    # held, and the two non-default gradient arithmetics (seeds beyond every earlier round's)
    G='^FAIL|cases passed|binning mode'
    timeout 900 python tools/fuzz_parity.py 800 20000 > $O/fz_sweep.txt 2>&1; grep -E "$G" $O/fz_sweep.txt | cut -c1-220
    timeout 900 python tools/fuzz_parity.py deep 400 9000 > $O/fz_deep.txt 2>&1; grep -E "$G" $O/fz_deep.txt | cut -c1-220
    timeout 700 python tools/fuzz_parity.py edge 500 10000 > $O/fz_edge.txt 2>&1; grep -E "$G" $O/fz_edge.txt | cut -c1-220
    GSR_PREPROCESS_AGG=1 timeout 700 python tools/fuzz_parity.py 400 21000 > $O/fz_sweep_agg.txt 2>&1; grep -E "$G" $O/fz_sweep_agg.txt | cut -c1-220
    GSR_PREPROCESS_AGG=1 GSR_FUZZ_BINS_KEYS=2048 timeout 700 python tools/fuzz_parity.py deep 250 9500 > $O/fz_deep_agg_2048.txt 2>&1; grep -E "$G" $O/fz_deep_agg_2048.txt | cut -c1-220
    GSR_TIERS_BESIDE_MAX=0 timeout 600 python tools/fuzz_parity.py deep 150 9800 > $O/fz_deep_not_held.txt 2>&1; grep -E "$G" $O/fz_deep_not_held.txt | cut -c1-220
    GSR_FUZZ_GRAD_PRECISION=accurate timeout 700 python tools/fuzz_parity.py 400 22000 > $O/fz_sweep_accurate.txt 2>&1; grep -E "$G" $O/fz_sweep_accurate.txt | cut -c1-220
    GSR_FUZZ_GRAD_PRECISION=accurate timeout 700 python tools/fuzz_parity.py edge 300 10600 > $O/fz_edge_accurate.txt 2>&1; grep -E "$G" $O/fz_edge_accurate.txt | cut -c1-220
    GSR_FUZZ_GRAD_PRECISION=fp32_reference timeout 700 python tools/fuzz_parity.py edge 300 10600 > $O/fz_edge_fp32ref.txt 2>&1; grep -E "$G" $O/fz_edge_fp32ref.txt | cut -c1-220
    GSR_FUZZ_GRAD_PRECISION=accurate timeout 600 python tools/fuzz_parity.py deep 150 9950 > $O/fz_deep_accurate.txt 2>&1; grep -E "$G" $O/fz_deep_accurate.txt | cut -c1-220
    timeout 300 python tools/fuzz_parity.py trainer 40 300 > $O/fz_trainer.txt 2>&1; grep -E "$G" $O/fz_trainer.txt | cut -c1-220
    timeout 300 python tools/fuzz_parity.py ssim 300 2000 > $O/fz_ssim.txt 2>&1; grep -E "$G" $O/fz_ssim.txt | cut -c1-220
    ;;
arbitrate)
    # every failing case of the campaign against the float64 autograd model (tools/fuzz_parity.py arbitrate: criteria (a) / (b) / (c))
    timeout 900 python tools/fuzz_parity.py arbitrate sweep 20093 20308 > $O/arb_sweep.txt 2>&1
    timeout 900 python tools/fuzz_parity.py arbitrate deep 9026 9030 9094 9163 9259 9394 > $O/arb_deep.txt 2>&1
    GSR_FUZZ_GRAD_PRECISION=accurate timeout 900 python tools/fuzz_parity.py arbitrate deep 9026 9030 9094 9163 9259 9394 > $O/arb_deep_accurate.txt 2>&1
    GSR_PREPROCESS_AGG=1 GSR_FUZZ_BINS_KEYS=2048 timeout 600 python tools/fuzz_parity.py arbitrate deep 9587 > $O/arb_deep_agg.txt 2>&1
    timeout 900 python tools/fuzz_parity.py arbitrate edge 10081 10098 10100 10101 10131 10172 10199 10275 10426 10439 > $O/arb_edge.txt 2>&1
    GSR_FUZZ_GRAD_PRECISION=accurate timeout 900 python tools/fuzz_parity.py arbitrate sweep 22028 22243 > $O/arb_sweep_accurate.txt 2>&1
    GSR_FUZZ_GRAD_PRECISION=accurate timeout 900 python tools/fuzz_parity.py arbitrate edge 10603 10625 10649 10652 10657 10674 10854 10873 > $O/arb_edge_accurate.txt 2>&1
    GSR_FUZZ_GRAD_PRECISION=fp32_reference timeout 900 python tools/fuzz_parity.py arbitrate edge 10603 10625 10649 10652 10657 10674 10854 10873 > $O/arb_edge_fp32ref.txt 2>&1
    timeout 300 python tools/fuzz_parity.py trainer 40 300 > $O/fz_trainer.txt 2>&1; grep -E "^FAIL|cases passed" $O/fz_trainer.txt | cut -c1-220
    tail -n 40 $O/arb_deep.txt $O/arb_deep_accurate.txt
    ;;
esac
