#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_ties_shell.py tests/test_gpu_mesh_build.py -q -x > gpurun_out/r4af_tests.log 2>&1; tail -3 gpurun_out/r4af_tests.log
REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2> gpurun_out/r4af_ad_err.log | grep "^adaptive" | cut -c1-32 | tr "\n" " "; echo
python tools/time_coarse.py 2>/dev/null | grep "^slab"
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4af_prof -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4af_prof.log 2>&1
cd $R
DB=$(find gpurun_out/r4af_prof -name "*.db" | head -1)
python tools/rocprof_summary.py timeline $DB 36 > gpurun_out/r4af_timeline.csv; grep "fused\|nearest" gpurun_out/r4af_timeline.csv
find gpurun_out -name "*.db" -delete
