#!/bin/bash
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4i_prof -- env WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4i_prof.log 2>&1
cd $R; grep -v "^/opt" gpurun_out/r4i_prof.log | tail -5
python tools/rocprof_summary.py stats $(find gpurun_out/r4i_prof -name "*.db" | head -1) > gpurun_out/r4i_kernel_stats.csv; head -45 gpurun_out/r4i_kernel_stats.csv
find gpurun_out -name "*.db" -delete
