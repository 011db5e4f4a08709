#!/bin/bash
# round 4, first GPU call: the device mesh build against its checker + what the refactor touched + a quick bench
mkdir -p gpurun_out
export ICON_AMD_DUMP_DIR=$PWD/gpurun_out/dump
timeout 600 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q -x 2>&1 | tail -40 > gpurun_out/r4a_meshbuild.log; tail -25 gpurun_out/r4a_meshbuild.log
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -k "hidden_activations or host_built or vertex_normals or sdf_query_vs_oracle or non_finite or extreme" 2>&1 | tail -15 > gpurun_out/r4a_parity_subset.log; tail -8 gpurun_out/r4a_parity_subset.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r4a_bench.log 2>&1; tail -1 gpurun_out/r4a_bench.log | cut -c1-1500
