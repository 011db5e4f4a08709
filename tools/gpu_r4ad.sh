#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r04_pytest.log 2>&1; tail -4 gpurun_out/r04_pytest.log
