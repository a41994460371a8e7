set -x
O=gpurun_out/r04m; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py arbitrate edge 25 43 51 77 205 257 301 316 358 383 384 395 406 418 420 559 569 576 622 728 803 880 897 931 957 1001 1155 1220 1263 1323 1342 > $O/arbitrate_edge.txt 2>&1; echo "rc=$?" >> $O/arbitrate_edge.txt
grep -c "NOT EXPLAINED" $O/arbitrate_edge.txt; tail -3 $O/arbitrate_edge.txt | cut -c1-300
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
GSR_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 3 2> $O/bench_1rank_rccl_torchrun.err | grep "^{" > $O/bench_1rank_rccl_torchrun.json
cut -c1-200 $O/bench_1rank_rccl_torchrun.json
