set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r01_g}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$R.log 2>&1
python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
python bench.py --no-cpu-baseline --reference-lists > gpurun_out/bench_${R}_reflists.json 2>/dev/null
python bench.py --no-cpu-baseline --gaussians 100000 --no-loss > gpurun_out/bench_${R}_cfg2.json 2>/dev/null
python bench.py --no-cpu-baseline --gaussians 5000000 --width 3840 --height 2160 --no-loss --steps 10 > gpurun_out/bench_${R}_cfg5.json 2>/dev/null
python bench.py --no-cpu-baseline --with-optimizer > gpurun_out/bench_${R}_optimizer.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o bench -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/prof_$R.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$R -o fetch -- python3 tools/pmc_workload.py > gpurun_out/pmc_${R}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_$R -o write -- python3 tools/pmc_workload.py > gpurun_out/pmc_${R}_write.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/sq_$R -o sq_pass1 -- python3 tools/pmc_workload.py > gpurun_out/sq_${R}_1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/sq_$R -o sq_pass2 -- python3 tools/pmc_workload.py > gpurun_out/sq_${R}_2.log 2>&1
tail -2 gpurun_out/smoke_$R.log; cat gpurun_out/bench_$R.json | cut -c1-400
