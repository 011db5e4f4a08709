#!/bin/bash
# the whole GPU suite on the box, result under gpurun_out/<tag>_pytest.log     usage: gpurun -- 'bash tools/gpu_suite.sh <tag>'
T=${1:-rX}
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; tail -4 gpurun_out/${T}_pytest.log
