#!/bin/bash
for pk in 4 2; do for s in 4 8 16 0; do echo "== packet $pk split $s"; ICON_AMD_PACKET=$pk ICON_AMD_SPLIT=$s REPEAT=3 WHICH=adaptive timeout 100 python tools/time_adaptive.py 2>&1 | grep "^adaptive" | cut -c1-32 | tr "\n" " "; echo; done; done
