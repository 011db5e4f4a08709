#!/bin/bash
mkdir -p gpurun_out
export ICON_AMD_SPLIT_TILES=100000
for cfg in "2 0" "4 0" "4 8" "4 16" "2 8"; do set -- $cfg; echo -n "packet $1 split $2: "; ICON_AMD_PACKET=$1 ICON_AMD_SPLIT=$2 timeout 100 python tools/time_coarse.py 2> gpurun_out/r4t_err_$1_$2.log | grep "^slab"; echo; done
