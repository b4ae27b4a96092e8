#!/bin/bash
# round 6: parity of the block path after a join change + the isolated timings on the three shapes + the kernel table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${TAG:-join}
timeout 1500 python -m pytest tests/test_gpu_block.py tests/test_gpu_fullsize.py tests/test_gpu_pp.py tests/test_gpu_frames.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r06_${TAG}_tests.txt
cat gpurun_out/r06_${TAG}_tests.txt
{
python tools/pp_block_probe.py --scans 32 --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block\|DIFFER'
python tools/pp_block_probe.py --scans 32 --reps 6 --shards 2 --matched 8,3,15 2>&1 | grep 'PARITY\|^block\|DIFFER'
python tools/pp_block_probe.py --scans 32 --reps 6 --shards 2 --nusc --trav 20 --frames 16 --n 35000 2>&1 | grep 'PARITY\|^block\|DIFFER'
python tools/pp_block_probe.py --scans 16 --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block\|DIFFER'
python tools/pp_block_probe.py --scans 4 --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block\|DIFFER'
} > gpurun_out/r06_${TAG}_probe.txt
cat gpurun_out/r06_${TAG}_probe.txt
PP_BLOCK=32 KS_NAME=r06_${TAG}_kernel_stats.csv bash tools/r05_kstats.sh 2>&1 | head -24
