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
esac
