#!/bin/bash
# Same-box A/B of the geometry pre-pass: the round-3 / round-4 trees (build_ab/r03, r04: `git archive 3a678e2 | 534525d | tar -x` + make) against
# this tree with and without the per-call tile record (ICON_AMD_LATTICE_FAST); kernel times from rocprofv3, interleaved twice.
#   prepare (once, in the build container; build_ab/ is git-ignored but travels with the gpurun snapshot):
#     for t in r03:3a678e2 r04:534525d; do d=build_ab/${t%%:*}; mkdir -p $d && git archive ${t##*:} | tar -x -C $d && make -j8 -C $d/icon_amd/csrc; done
#   usage: gpurun -- 'bash tools/ab_r05.sh <tag>'        (profiles/r05_ab_nearest.txt is one such session)
T=${1:-ab5}
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench path, env...
  n=$1; b=$2; shift 2
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${T}_$n -- python $b --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${T}_$n.log 2>&1
  python $R/tools/rocprof_summary.py stats $(find $R/gpurun_out/${T}_$n -name "*.db" | head -1) > $R/gpurun_out/${T}_${n}_stats.csv
  echo "== $n"; grep "k_nearest\|k_fused\|k_sign\|k_row" $R/gpurun_out/${T}_${n}_stats.csv | cut -c1-150
  find $R/gpurun_out/${T}_$n -name "*.db" -delete
}
for rep in 1 2; do
  [ -d $R/build_ab/r03 ] && run r03_$rep $R/build_ab/r03/bench.py A=1
  [ -d $R/build_ab/r04 ] && run r04_$rep $R/build_ab/r04/bench.py A=1
  run slow_$rep $R/bench.py ICON_AMD_LATTICE_FAST=0
  run fast_$rep $R/bench.py ICON_AMD_LATTICE_FAST=1
done
