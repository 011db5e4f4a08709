#!/bin/bash
set -x
mkdir -p gpurun_out
AMD_LOG_LEVEL=1 timeout 300 python tools/dbg_seg5.py 5 > gpurun_out/r2d_seg5.log 2>&1; tail -25 gpurun_out/r2d_seg5.log
timeout 300 python tools/dbg_seg5.py 4 2>&1 | tail -3
timeout 600 python -X faulthandler -m pytest tests/test_mesh_tools.py -m gpu -x -q 2>&1 | tail -15
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or pamir_lattice or 257 or 513 or zslab or dense_recon or adaptive or chamfer" 2>&1 | tail -15
