#!/bin/bash
mkdir -p gpurun_out
WHICH=adaptive timeout 100 python tools/time_adaptive.py 2>&1 | grep "^adaptive" | cut -c1-200
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4n_prof -- env WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4n_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4n_prof -name "*.db" | head -1) > gpurun_out/r4n_kernel_stats.csv; head -8 gpurun_out/r4n_kernel_stats.csv
find gpurun_out -name "*.db" -delete
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_ties_shell.py -q -k "adaptive or native_schedule or lattice_vs_oracle or shell_skip or random_lattices" 2>&1 | tail -3
