#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_ties_shell.py -q -x -k "native_schedule or adaptive or lattice or shared_walk or rows_entry or query" > gpurun_out/r4x_tests.log 2>&1; tail -3 gpurun_out/r4x_tests.log
echo -n "default: "; REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2> gpurun_out/r4x_ad_err.log | grep "^adaptive" | cut -c1-32 | tr "\n" " "; echo
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4x_prof -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4x_prof.log 2>&1
cd $R
DB=$(find gpurun_out/r4x_prof -name "*.db" | head -1)
python tools/rocprof_summary.py stats $DB > gpurun_out/r4x_kernel_stats.csv
python tools/rocprof_summary.py timeline $DB 38 > gpurun_out/r4x_timeline.csv; cat gpurun_out/r4x_timeline.csv
find gpurun_out -name "*.db" -delete
