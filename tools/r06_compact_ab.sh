#!/bin/bash
# round 6: packing of a task's member records (MODEST_PP4_COMPACT=1, the default) against none, on the four shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06_compact_ab.txt
for args in "" "--matched 8,3,15" "--matched 8,3,15 --presence 5" "--nusc --trav 20 --frames 16 --n 35000" "--matched 5,3,5"; do
  for c in 1 0 1 0; do
    echo "[$args] compact=$c: $(MODEST_PP4_COMPACT=$c python tools/pp_block_probe.py --scans 32 --reps 5 --shards 2 $args 2>&1 | grep 'PARITY\|^block\|DIFFER' | tr '\n' ' ')" >> gpurun_out/r06_compact_ab.txt
  done
done
cat gpurun_out/r06_compact_ab.txt
