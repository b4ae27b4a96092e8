#!/bin/bash
# segment size of the block store's counting sort (a scan reads whole segments: the boundary segments carry other scans' frames)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/b5.txt
for seg in 2048 1024 4096; do
  MODEST_EXTRA_CXXFLAGS="-DB4_SEG_=$seg" python -c "from modest_amd import build; build.build(verbose=False)" > gpurun_out/b5_build_$seg.log 2>&1
  export MODEST_EXTRA_CXXFLAGS="-DB4_SEG_=$seg"
  for g in 16 4; do
    echo "SEG $seg scans $g: $(MODEST_PP4_TK=2 python tools/pp_block_probe.py --scans $g --reps 6 --shards 2 2>&1 | grep 'PARITY\|^block' | tr '\n' ' ')" >> gpurun_out/b5.txt
  done
  PP_BLOCK=16 KS_NAME=b5_ks_$seg.csv bash tools/r05_kstats.sh 2>&1 | grep "b4_" | head -12 >> gpurun_out/b5.txt
  unset MODEST_EXTRA_CXXFLAGS
done
cat gpurun_out/b5.txt
