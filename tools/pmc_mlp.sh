#!/bin/bash
# two rocprofv3 --pmc passes over the MLP kernel alone; run on the GPU box: tools/pmc_mlp.sh <precision>
P=${1:-f16x3}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmc_${P}_a -- python $R/tools/mlp_only.py $P 3 > $R/gpurun_out/pmc_${P}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES -d $R/gpurun_out/pmc_${P}_b -- python $R/tools/mlp_only.py $P 3 > $R/gpurun_out/pmc_${P}_b.log 2>&1
cd $R
for d in gpurun_out/pmc_${P}_a gpurun_out/pmc_${P}_b; do f=$(find $d -name "*.db" | head -1); python tools/pmc_extract.py $f | grep -A12 "k_mlp"; done
tail -1 gpurun_out/pmc_${P}_a.log
