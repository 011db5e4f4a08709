#!/bin/bash
mkdir -p gpurun_out
for pk in auto 1 2 4; do
  if [ $pk = auto ]; then unset ICON_AMD_PACKET; else export ICON_AMD_PACKET=$pk; fi
  echo "== packet $pk"; WHICH=adaptive timeout 100 python tools/time_adaptive.py 2>&1 | grep "^adaptive" | cut -c1-200
done
unset ICON_AMD_PACKET
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4j_prof -- env WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4j_prof.log 2>&1
cd $R
python tools/rocprof_summary.py stats $(find gpurun_out/r4j_prof -name "*.db" | head -1) > gpurun_out/r4j_kernel_stats.csv; head -14 gpurun_out/r4j_kernel_stats.csv
find gpurun_out -name "*.db" -delete
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -k "adaptive or native_schedule or lattice_vs_oracle or shell_skip" 2>&1 | tail -4
