#!/bin/bash
mkdir -p gpurun_out
for k in 1 2 3 4 5 6; do timeout 120 python -X faulthandler tools/stress_adaptive.py > gpurun_out/r4y_stress_$k.log 2>&1; echo "run $k rc=$? $(grep -a 'stress ok\|MISMATCH\|fault\|rror' gpurun_out/r4y_stress_$k.log | head -3)"; done
