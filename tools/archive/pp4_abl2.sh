#!/bin/bash
cd $GRAFT_REPO_ROOT
for sc in 1 2 4 8; do
 for d in 0 64 32; do
  echo "== scans $sc dbg $d"; MODEST_PP_BLOCK=1 MODEST_PP4_DBG=$d python tools/pp_block_probe.py --scans $sc --reps 4 --shards 2 2>&1 | grep "^block"
 done
done
