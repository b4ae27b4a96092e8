#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_block.py -x -q 2>&1 | tail -1) > gpurun_out/b4.txt
for tk in 1 2 3; do
  for g in 16 4; do
    echo "TK $tk scans $g help: $(MODEST_PP4_TK=$tk python tools/pp_block_probe.py --scans $g --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block' | tr '\n' ' ')" >> gpurun_out/b4.txt
    echo "TK $tk scans $g nohelp: $(MODEST_PP4_DBG=1024 MODEST_PP4_TK=$tk python tools/pp_block_probe.py --scans $g --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block' | tr '\n' ' ')" >> gpurun_out/b4.txt
  done
done
MODEST_PP4_TK=2 MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 16 --reps 1 --shards 1 2>&1 | grep "b4_join" | head -20 >> gpurun_out/b4.txt
cat gpurun_out/b4.txt
