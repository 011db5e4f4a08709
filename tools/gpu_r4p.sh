#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q -x 2>&1 | tail -3
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_ties_shell.py -q -x -k "adaptive or native_schedule or lattice_vs_oracle or shell_skip or random_lattices" 2>&1 | tail -3
for s in 1 0; do echo "== split $s"; ICON_AMD_SPLIT=$s WHICH=adaptive timeout 100 python tools/time_adaptive.py 2>&1 | grep "^adaptive" | cut -c1-200; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4p_prof -- env WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4p_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4p_prof -name "*.db" | head -1) > gpurun_out/r4p_kernel_stats.csv; head -12 gpurun_out/r4p_kernel_stats.csv | cut -c1-110
find gpurun_out -name "*.db" -delete
