#!/bin/bash
for pk in auto 1 4; do
  if [ $pk = auto ]; then unset ICON_AMD_PACKET; else export ICON_AMD_PACKET=$pk; fi
  echo "== packet $pk"; timeout 100 python tools/trav_stats.py 2>&1 | grep "^33\|^65\|^129\|^257\|sdf_query" | cut -c1-200
done
unset ICON_AMD_PACKET
timeout 200 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -k "voxel_units or operands_beyond" 2>&1 | tail -3
