#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -x > gpurun_out/r4w_tests.log 2>&1; tail -4 gpurun_out/r4w_tests.log
for nw in 4 8 16; do echo -n "share $nw: "; ICON_AMD_SHARE=$nw REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2> gpurun_out/r4w_ad_err.log | grep "^adaptive" | cut -c1-32 | tr "\n" " "; echo; done
echo -n "default: "; REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2> gpurun_out/r4w_ad_err.log | grep "^adaptive" | cut -c1-32 | tr "\n" " "; echo
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4w_prof -- env REPEAT=2 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4w_prof.log 2>&1
cd $R
DB=$(find gpurun_out/r4w_prof -name "*.db" | head -1)
python tools/rocprof_summary.py stats $DB > gpurun_out/r4w_kernel_stats.csv
python tools/rocprof_summary.py timeline $DB 45 > gpurun_out/r4w_timeline.csv; cat gpurun_out/r4w_timeline.csv
find gpurun_out -name "*.db" -delete
