#!/bin/bash
cd $GRAFT_REPO_ROOT
for sc in ${GS:-4 8 12 16}; do
  echo "== scans $sc"; python tools/pp_block_probe.py --scans $sc --reps 3 --shards 2 2>&1 | grep "PARITY\|^block\|^chain"
done
