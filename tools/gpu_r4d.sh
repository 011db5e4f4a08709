#!/bin/bash
mkdir -p gpurun_out
export ICON_AMD_DUMP_DIR=$PWD/gpurun_out/dump
ICON_AMD_DEBUG_SYNC=1 timeout 100 python tools/time_mesh_build.py 2 > gpurun_out/r4d_time.log 2>&1; grep -v "^/opt" gpurun_out/r4d_time.log | tail -26
timeout 100 python tools/time_mesh_build.py 4 2>&1 | grep "^build"
timeout 200 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q 2>&1 | tail -30 > gpurun_out/r4d_meshbuild.log; grep -E "passed|failed|differs|Error|mismatch" gpurun_out/r4d_meshbuild.log | head
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -k "hidden_activations or host_built or vertex_normals or sdf_query_vs_oracle or non_finite or extreme" 2>&1 | tail -15 > gpurun_out/r4d_parity_subset.log; tail -4 gpurun_out/r4d_parity_subset.log
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/r4d_bench.log 2>&1; tail -1 gpurun_out/r4d_bench.log | cut -c1-2500
