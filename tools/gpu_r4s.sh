#!/bin/bash
run() { echo "== mode $1 packet $2 split $3"; ICON_AMD_SPLIT_MODE=$1 ICON_AMD_PACKET=$2 ICON_AMD_SPLIT=$3 timeout 150 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -k "native_schedule or lattice_vs_oracle" 2>&1 | grep -v "^W2026\|^E2026\|amdgpu.ids" | tail -${4:-2} | cut -c1-250; }
run 1 4 8 30
run 1 4 8 30
run 1 4 16 30
run 1 2 8 30
export ICON_AMD_SPLIT_MODE=1
for pk in 4 2; do for s in 4 8 16; do echo "== packet $pk split $s"; ICON_AMD_PACKET=$pk ICON_AMD_SPLIT=$s REPEAT=3 WHICH=adaptive timeout 60 python tools/time_adaptive.py 2>&1 | grep "^adaptive\|rror\|fault" | cut -c1-32 | tr "\n" " "; echo; done; done
