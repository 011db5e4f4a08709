#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -k "native_schedule or adaptive" > gpurun_out/r4v_tests.log 2>&1; tail -3 gpurun_out/r4v_tests.log
REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2> gpurun_out/r4v_ad_err.log | grep "^adaptive" | cut -c1-90
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4v_prof -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4v_prof.log 2>&1
cd $R
DB=$(find gpurun_out/r4v_prof -name "*.db" | head -1)
python tools/rocprof_summary.py stats $DB > gpurun_out/r4v_kernel_stats.csv; head -24 gpurun_out/r4v_kernel_stats.csv | cut -c1-110
python tools/rocprof_summary.py timeline $DB 60 > gpurun_out/r4v_timeline.csv; cat gpurun_out/r4v_timeline.csv
find gpurun_out -name "*.db" -delete
