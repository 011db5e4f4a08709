#!/bin/bash
PROFILES_ONLY=1 bash tools/gpu_round.sh r04 2>&1 | tail -30
cp gpurun_out/r04_traffic.json profiles/traffic.json
timeout 600 python bench.py > gpurun_out/r04_bench.log 2>&1; tail -1 gpurun_out/r04_bench.log | cut -c1-200
