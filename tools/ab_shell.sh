#!/bin/bash
# A/B on one box: kernel times of the 257^3 step with and without the shell skip (rocprofv3 kernel stats)
R=$PWD; T=${1:-ab}
cd /tmp && export TMPDIR=/tmp
for s in 1 0; do
  ICON_AMD_SHELL_SKIP=$s timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_skip$s -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_skip$s.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/${T}_skip$s -name "*.db" | head -1) > $R/gpurun_out/${T}_skip${s}_stats.csv
  echo "== shell skip $s"; head -8 $R/gpurun_out/${T}_skip${s}_stats.csv
done
find $R/gpurun_out -name "*.db" -delete
