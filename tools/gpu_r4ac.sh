#!/bin/bash
for m in default nogc nosync dense; do MODE=$m timeout 60 python tools/probe_stall.py 2>/dev/null | cut -c1-400; done
