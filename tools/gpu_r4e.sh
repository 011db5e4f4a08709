#!/bin/bash
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4e_prof -- python $R/tools/time_mesh_build.py 6 > $R/gpurun_out/r4e_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4e_prof -name "*.db" | head -1) > gpurun_out/r4e_kernel_stats.csv; head -30 gpurun_out/r4e_kernel_stats.csv
find gpurun_out -name "*.db" -delete
