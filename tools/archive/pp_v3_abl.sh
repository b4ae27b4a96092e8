#!/bin/bash
# PP ablations on the GPU box: MODEST_PP_DBG bits (pp_v3.h): 1 no counter flush, 2 no candidate loops,
# 4 no bands at all, 8 in-kernel phase timers (stderr), 16 scatter without stores, 64 count without atomics.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MODEST_PP_DBG=8 timeout 300 python tools/pp_microbench.py 2>&1 | grep "pp3" | tail -3 | cut -c1-400
for d in 0 1 2 4; do echo "== dbg $d"; MODEST_PP_DBG=$d timeout 300 python tools/pp_microbench.py 2>&1 | grep "ms_per_scan" | tail -1 | cut -c1-70; done
