#!/bin/bash
# A/B on one box: k_nearest with / without the per-row rotation of the block -> x mapping (and without the shell skip)
R=$PWD; T=${1:-abn}
cd /tmp && export TMPDIR=/tmp
for cfg in "1 0" "1 3" "0 0"; do
  set -- $cfg
  ICON_AMD_SHELL_SKIP=$1 ICON_AMD_XCD_REMAP=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_$1_$2 -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_$1_$2.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/${T}_$1_$2 -name "*.db" | head -1) > $R/gpurun_out/${T}_$1_$2_stats.csv
  echo "== shell skip $1 remap $2"; head -4 $R/gpurun_out/${T}_$1_$2_stats.csv
done
find $R/gpurun_out -name "*.db" -delete
