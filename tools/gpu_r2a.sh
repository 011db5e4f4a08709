#!/bin/bash
# round 2, call a: full GPU test suite, the default bench line (f16x3 + extras), PMC of the default MLP kernel
set -x
R=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a_pytest.log; tail -5 gpurun_out/r2a_pytest.log
timeout 600 python bench.py > gpurun_out/r2a_bench.log 2>&1; tail -1 gpurun_out/r2a_bench.log | cut -c1-3000
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r2a_pmc_a -- python $R/tools/mlp_only.py f16x3 3 > $R/gpurun_out/r2a_pmc_a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/r2a_pmc_c -- python $R/tools/mlp_only.py f16x3 3 > $R/gpurun_out/r2a_pmc_c.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2a_stats -- python $R/tools/mlp_only.py f16x3 3 > $R/gpurun_out/r2a_stats.log 2>&1
cd $R
for d in gpurun_out/r2a_pmc_a gpurun_out/r2a_pmc_c; do f=$(find $d -name "*.db" | head -1); python tools/pmc_extract.py $f | grep -A10 "k_mlp"; done > gpurun_out/r2a_pmc.txt 2>&1
cat gpurun_out/r2a_pmc.txt
find gpurun_out/r2a_stats -name "*kernel_stats.csv" -exec head -5 {} \;
find gpurun_out -name "*.db" -size +20M -delete
