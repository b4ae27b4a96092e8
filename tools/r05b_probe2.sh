#!/bin/bash
# GPU box: block parity tests + probe timings (16, 4 scans) + the join's wavefront end times
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_block.py -x -q 2>&1 | tail -3) > gpurun_out/b2_tests.txt
python tools/pp_block_probe.py --scans 16 --reps 6 --shards 2 2>&1 | grep "PARITY\|^block\|^chain\|DIFFER" > gpurun_out/b2_p16.txt
python tools/pp_block_probe.py --scans 4 --reps 6 --shards 2 2>&1 | grep "PARITY\|^block\|^chain\|DIFFER" > gpurun_out/b2_p4.txt
MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 16 --reps 1 --shards 1 2>&1 | grep "b4_join" | grep -v "scan [0-9]" | head -2 > gpurun_out/b2_tail16.txt
MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 4 --reps 1 --shards 1 2>&1 | grep "b4_join" | grep -v "scan [0-9]" | head -2 > gpurun_out/b2_tail4.txt
PP_BLOCK=16 KS_NAME=b2_ks16.csv bash tools/r05_kstats.sh > gpurun_out/b2_ks16.txt 2>&1
tail -n 3 gpurun_out/b2_tests.txt; cat gpurun_out/b2_p16.txt gpurun_out/b2_p4.txt gpurun_out/b2_tail16.txt gpurun_out/b2_tail4.txt; head -5 gpurun_out/b2_ks16.txt
