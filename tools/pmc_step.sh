#!/bin/bash
# PMC passes over one full bench step (all kernels); run on the GPU box: tools/pmc_step.sh
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_step_a -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_step_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_SMEM SQ_WAVES SQ_INSTS_BRANCH -d $R/gpurun_out/pmc_step_b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_step_b.log 2>&1
cd $R
for d in gpurun_out/pmc_step_a gpurun_out/pmc_step_b; do f=$(find $d -name "*.db" | head -1); python tools/pmc_extract.py $f | grep -A9 -E "k_nearest|k_features<"; done
