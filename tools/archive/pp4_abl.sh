#!/bin/bash
# ablation of b4_join: MODEST_PP4_DBG bits 1 no pair loop (heavy), 2 no light path, 4 no pose gather, 8 no heavy path
cd $GRAFT_REPO_ROOT
python tools/pp_block_probe.py --scans 8 --reps 4 --shards 2 2>&1 | grep "PARITY\|^block\|^chain\|DIFFER"
for d in ${DBGS:-1 2 4 8 10}; do
  echo "== dbg $d"; MODEST_PP4_DBG=$d python tools/pp_block_probe.py --scans 8 --reps 4 --shards 2 2>&1 | grep "^block"
done
for j in ${JWGS:-2 6}; do
  echo "== jwg $j"; MODEST_PP4_JWG=$j python tools/pp_block_probe.py --scans 8 --reps 4 --shards 2 2>&1 | grep "^block"
done
