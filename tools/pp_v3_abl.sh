#!/bin/bash
for sr in 4096 2048 1024; do echo "== slice $sr"; MODEST_PP_SLICE=$sr timeout 300 python tools/pp_microbench.py 2>&1 | grep "ms_per_scan" | tail -1 | cut -c1-70; done
