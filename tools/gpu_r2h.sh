#!/bin/bash
# A/B of the fused kernel bodies under PMC: cycles vs time (is the kernel clock/power bound?)
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
ICON_AMD_FUSED_VER=$v timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r2h_pmc_v$v -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2h_pmc_v$v.log 2>&1
ICON_AMD_FUSED_VER=$v timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2h_stats_v$v -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2h_stats_v$v.log 2>&1
done
cd $R
for v in 1 2; do echo "== ver $v"; f=$(find gpurun_out/r2h_pmc_v$v -name "*.db" | head -1); python tools/pmc_extract.py $f | grep -A9 "k_fused"; f=$(find gpurun_out/r2h_stats_v$v -name "*.db" | head -1); python tools/rocprof_summary.py stats $f | head -3; done
find gpurun_out -name "*.db" -size +20M -delete
