#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -X faulthandler -m pytest tests/test_gpu_mesh_build.py -q -x > gpurun_out/r4ae_tests.log 2>&1; tail -3 gpurun_out/r4ae_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r04_bench.log 2>&1; tail -1 gpurun_out/r04_bench.log | cut -c1-300
