#!/bin/bash
# round 2, call c: fused path after the spill fixes - full test log, bench, kernel stats, HBM traffic
set -x
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -v > gpurun_out/r2c_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r2c_pytest.log | tail -3; grep -B2 -A12 "Fatal\|Aborted\|Memory access fault\|FAILED" gpurun_out/r2c_pytest.log | head -60
timeout 600 python bench.py > gpurun_out/r2c_bench.log 2>&1; tail -1 gpurun_out/r2c_bench.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2c_stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2c_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r2c_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2c_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r2c_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2c_write.log 2>&1
cd $R
for d in gpurun_out/r2c_fetch gpurun_out/r2c_write; do f=$(find $d -name "*.db" | head -1); python tools/pmc_extract.py $f; done > gpurun_out/r2c_traffic.txt 2>&1
grep -v "^counters" gpurun_out/r2c_traffic.txt
find gpurun_out/r2c_stats -name "*kernel_stats.csv" -exec cat {} \; | head -14
find gpurun_out -name "*.db" -size +20M -delete
