#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -X faulthandler -m pytest tests/test_gpu_ties_shell.py tests/test_gpu_parity.py tests/test_gpu_mesh_build.py -q -x -k "shared_walk or native_schedule or lattice or adaptive or rows_entry or mesh_build or device_build" > gpurun_out/r4u_tests.log 2>&1; tail -3 gpurun_out/r4u_tests.log
timeout 100 python tools/time_coarse.py 2> gpurun_out/r4u_coarse_err.log | grep "^slab"
REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2> gpurun_out/r4u_ad_err.log | grep "^adaptive" | cut -c1-90
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4u_prof -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4u_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4u_prof -name "*.db" | head -1) > gpurun_out/r4u_kernel_stats.csv; head -40 gpurun_out/r4u_kernel_stats.csv | cut -c1-110
find gpurun_out -name "*.db" -delete
