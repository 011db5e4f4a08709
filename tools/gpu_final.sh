#!/bin/bash
# the end-of-round call: whole GPU suite, the profile legs of tools/gpu_round.sh, the default bench line with this tree's traffic.json
T=${1:-rX}
bash tools/gpu_suite.sh $T
PROFILES_ONLY=1 bash tools/gpu_round.sh $T 2>&1 | tail -14
cp gpurun_out/${T}_traffic.json profiles/traffic.json
timeout 600 python bench.py > gpurun_out/${T}_bench.log 2>&1; tail -1 gpurun_out/${T}_bench.log | cut -c1-200
timeout 100 python tools/time_mesh_extract.py 2>/dev/null | grep "^schedule" > gpurun_out/${T}_mesh_extract.txt; cat gpurun_out/${T}_mesh_extract.txt
