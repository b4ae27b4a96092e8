#!/bin/bash
# the block path on the other shapes: C5 (nuScenes), reference-rule windows by speed bucket
cd $GRAFT_REPO_ROOT
echo "== C5"; python tools/pp_block_probe.py --scans 16 --reps 3 --shards 2 --n 35000 --trav 20 --frames 16 --nusc 2>&1 | grep "PARITY\|^block\|^chain\|DIFFER"
for m in 3,10,15 8,5,10 8,3,15 5,3,5 12,5,10; do
  echo "== matched $m"; python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 --matched $m 2>&1 | grep "sharing\|PARITY\|^block\|^chain\|DIFFER"
done
echo "== matched nusc 8,3,15"; python tools/pp_block_probe.py --scans 16 --reps 3 --shards 1 --n 35000 --trav 20 --nusc --matched 8,3,15 2>&1 | grep "sharing\|PARITY\|^block\|^chain\|DIFFER"
