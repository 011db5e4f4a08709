#!/bin/bash
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM -d $R/gpurun_out/r4aa_pmc -- env REPEAT=1 WHICH=adaptive python $R/tools/time_adaptive.py > $R/gpurun_out/r4aa_pmc.log 2>&1
cd $R
python tools/pmc_extract.py $(find gpurun_out/r4aa_pmc -name "*.db" | head -1) > gpurun_out/r4aa_pmc.txt; grep -A9 "k_ad_nearest\|k_nearest_shared\|k_sign_wide" gpurun_out/r4aa_pmc.txt | head -60
find gpurun_out -name "*.db" -delete
python tools/trav_stats.py 2>&1 | grep "^33\|^65\|^129\|^257" | cut -c1-250
