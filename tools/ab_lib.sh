#!/bin/bash
# A/B on one box: kernel times of the 257^3 step under different builds of the library (ICON_AMD_LIB); interleaved twice
#   usage: tools/ab_lib.sh <tag> <name=path-to-lib | name=> ...     (empty path = the in-tree library)
R=$PWD; T=$1; shift
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for nv in "$@"; do
    n=${nv%%=*}; lib=${nv#*=}
    if [ -n "$lib" ]; then export ICON_AMD_LIB=$R/$lib; else unset ICON_AMD_LIB; fi
    timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_${n}_$rep -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_${n}_$rep.log 2>&1
    python $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/${T}_${n}_$rep -name "*.db" | head -1) > $R/gpurun_out/${T}_${n}_${rep}_stats.csv
    echo "== $n (run $rep)"; head -3 $R/gpurun_out/${T}_${n}_${rep}_stats.csv | tail -2
  done
done
find $R/gpurun_out -name "*.db" -delete
