#!/bin/bash
mkdir -p gpurun_out
export ICON_AMD_PACKET=4 ICON_AMD_SPLIT=8
PERCALL=1 WHICH=none timeout 100 python tools/time_adaptive.py 2>&1 | grep "per call" | cut -c1-400
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4r_prof -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4r_prof.log 2>&1
cd $R; grep "^adaptive" gpurun_out/r4r_prof.log | cut -c1-40
python tools/rocprof_summary.py stats $(find gpurun_out/r4r_prof -name "*.db" | head -1) > gpurun_out/r4r_kernel_stats.csv; head -14 gpurun_out/r4r_kernel_stats.csv | cut -c1-110
find gpurun_out -name "*.db" -delete
