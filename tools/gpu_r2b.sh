#!/bin/bash
# round 2, call b: fused path - tests, bench line, kernel stats, HBM traffic (FETCH_SIZE / WRITE_SIZE passes)
set -x
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2b_pytest.log; tail -5 gpurun_out/r2b_pytest.log
timeout 600 python bench.py > gpurun_out/r2b_bench.log 2>&1; tail -1 gpurun_out/r2b_bench.log | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2b_stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2b_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2b_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2b_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2b_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2b_write.log 2>&1
cd $R
for d in gpurun_out/r2b_fetch gpurun_out/r2b_write; do f=$(find $d -name "*.db" | head -1); python tools/pmc_extract.py $f; done > gpurun_out/r2b_traffic.txt 2>&1
cat gpurun_out/r2b_traffic.txt | grep -v "^counters"
find gpurun_out/r2b_stats -name "*kernel_stats.csv" -exec cat {} \; | head -20
find gpurun_out -name "*.db" -size +20M -delete
