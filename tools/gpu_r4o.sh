#!/bin/bash
for sd in 1 2; do
for pk in auto 1; do
  if [ $pk = auto ]; then unset ICON_AMD_PACKET; else export ICON_AMD_PACKET=$pk; fi
  echo "== seeded $sd packet $pk"; ICON_AMD_STATS_SEEDED=$sd timeout 100 python tools/trav_stats.py 2>&1 | grep "^33\|^65\|^129\|^257" | cut -c1-260
done; done
