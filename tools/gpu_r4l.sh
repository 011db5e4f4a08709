#!/bin/bash
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4l_prof -- env WHICH=host python $R/tools/time_adaptive.py > $R/gpurun_out/r4l_prof.log 2>&1
cd $R; grep "^host" gpurun_out/r4l_prof.log | cut -c1-150
python tools/rocprof_summary.py stats $(find gpurun_out/r4l_prof -name "*.db" | head -1) > gpurun_out/r4l_kernel_stats.csv; head -12 gpurun_out/r4l_kernel_stats.csv
find gpurun_out -name "*.db" -delete
