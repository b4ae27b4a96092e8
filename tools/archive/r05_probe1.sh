#!/bin/bash
# round 5, first measurements: the block path as it stands (ablations of the join) and on reference-rule windows
cd $GRAFT_REPO_ROOT
python tools/pp_block_probe.py --scans 16 --reps 4 --shards 2 2>&1 | grep "PARITY\|^block\|^chain\|DIFFER"
for d in 1 2 8 10 512; do
  echo "== dbg $d"; MODEST_PP4_DBG=$d python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 2>&1 | grep "^block\|b4_join"
done
for m in 3,10,15 8,5,10 8,3,15 5,3,5; do
  echo "== matched $m"; python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 --matched $m 2>&1 | grep "sharing\|PARITY\|^block\|^chain\|DIFFER"
done
