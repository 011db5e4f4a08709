#!/bin/bash
mkdir -p gpurun_out
export ICON_AMD_DUMP_DIR=$PWD/gpurun_out/dump
timeout 100 python tools/time_mesh_build.py 4 2>&1 | grep "^build"
timeout 200 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q 2>&1 | tail -30 > gpurun_out/r4g_meshbuild.log; grep -E "passed|failed|differs|Error|mismatch" gpurun_out/r4g_meshbuild.log | head
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4g_prof -- python $R/tools/time_mesh_build.py 6 > $R/gpurun_out/r4g_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4g_prof -name "*.db" | head -1) > gpurun_out/r4g_kernel_stats.csv; head -16 gpurun_out/r4g_kernel_stats.csv
find gpurun_out -name "*.db" -delete
