#!/bin/bash
for g in 1 2 3 4 8; do for m in 64 256; do d=$(( (g<<8) | (m<<16) )); echo "== groups>=$g laneMax $m"; MODEST_PP_DBG=$d timeout 300 python tools/pp_microbench.py 2>&1 | grep "ms_per_scan" | tail -1 | cut -c1-60; done; done
