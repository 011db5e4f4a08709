#!/bin/bash
set -x
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r2f_pytest.log 2>&1; tail -4 gpurun_out/r2f_pytest.log
timeout 600 python bench.py > gpurun_out/r2f_bench.log 2>&1; tail -1 gpurun_out/r2f_bench.log | cut -c1-700
timeout 600 python bench.py --prior pamir --no-cpu-baseline > gpurun_out/r2f_bench_pamir.log 2>&1; tail -1 gpurun_out/r2f_bench_pamir.log | cut -c1-900
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2f_stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2f_stats.log 2>&1
cd $R
find gpurun_out/r2f_stats -name "*kernel_stats.csv" -exec cat {} \; | head -14
find gpurun_out -name "*.db" -size +20M -delete
