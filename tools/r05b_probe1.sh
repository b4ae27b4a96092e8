#!/bin/bash
# GPU box: gpu tests of the restored tree, the join's wavefront end-time statistics, kernel tables of blocks of 4 and 16
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/b1_tests.txt
MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 16 --reps 1 --shards 1 2>&1 | grep "b4_join\|PARITY\|^block" > gpurun_out/b1_join_tail16.txt
MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 4 --reps 1 --shards 1 2>&1 | grep "b4_join\|PARITY\|^block" > gpurun_out/b1_join_tail4.txt
PP_BLOCK=4 KS_NAME=b1_ks4.csv bash tools/r05_kstats.sh > gpurun_out/b1_ks4.txt 2>&1
PP_BLOCK=16 KS_NAME=b1_ks16.csv bash tools/r05_kstats.sh > gpurun_out/b1_ks16.txt 2>&1
cat gpurun_out/b1_tests.txt gpurun_out/b1_join_tail16.txt
