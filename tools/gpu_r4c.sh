#!/bin/bash
mkdir -p gpurun_out
export ICON_AMD_DUMP_DIR=$PWD/gpurun_out/dump
ICON_AMD_DEBUG_SYNC=1 timeout 200 python tools/time_mesh_build.py 2 > gpurun_out/r4c_time.log 2>&1; grep -v "^/opt" gpurun_out/r4c_time.log | tail -45
timeout 150 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q -k "equals_host and (ico or dup or tiny)" 2>&1 | tail -30 > gpurun_out/r4c_meshbuild.log; grep -E "passed|failed|differs|Error" gpurun_out/r4c_meshbuild.log | head
