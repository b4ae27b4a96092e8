#!/bin/bash
MODEST_PP_DBG=8 timeout 300 python tools/pp_microbench.py 2>&1 | grep "pp3" | tail -1
for d in 0 1 2 4; do echo "== dbg $d"; MODEST_PP_DBG=$d timeout 300 python tools/pp_microbench.py 2>&1 | grep "ms_per_scan" | tail -1; done
