set -x
O=gpurun_out/r04_fuzz; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py 2500 12 > $O/sweep.txt 2>&1; echo "rc=$?" >> $O/sweep.txt
timeout 1200 python tools/fuzz_parity.py deep 800 0 > $O/deep.txt 2>&1; echo "rc=$?" >> $O/deep.txt
timeout 1200 python tools/fuzz_parity.py edge 1500 0 > $O/edge.txt 2>&1; echo "rc=$?" >> $O/edge.txt
timeout 600 python tools/fuzz_parity.py ssim 300 0 > $O/ssim.txt 2>&1; echo "rc=$?" >> $O/ssim.txt
timeout 600 python tools/fuzz_parity.py trainer 40 0 > $O/trainer.txt 2>&1; echo "rc=$?" >> $O/trainer.txt
tail -3 $O/sweep.txt $O/deep.txt $O/edge.txt $O/ssim.txt $O/trainer.txt
