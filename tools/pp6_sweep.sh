#!/bin/bash
# gather-wave: class sizes and the sensitivity to the item-size knobs (GPU box)
cd $GRAFT_REPO_ROOT
export MODEST_PP_FRAMES_PATH=gather-wave
NSCAN=1 MODEST_PP5_PROF=1 timeout 120 python tools/pp5_microbench.py 2>&1 | grep "^\[pp5\]" | head -2
for cfg in "6144 256 1536 5" "3072 100000 1536 5" "6144 100000 1536 5" "12288 100000 1536 5" "6144 100000 768 5" "6144 100000 3072 5"; do
  set -- $cfg
  echo "== pmax $1 heavy $2 wmax $3 wgs $4"
  cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
  rm -rf gpurun_out/prof_sw
  MODEST_PP5_PMAX=$1 MODEST_PP5_HEAVY=$2 MODEST_PP6_WMAX=$3 MODEST_PP6_WGS=$4 NSCAN=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sw -o b -- python tools/pp5_microbench.py > gpurun_out/prof_sw.log 2>&1
  python tools/kstats.py gpurun_out/prof_sw/b_kernel_stats.csv 4 pp6_wave_join | head -4 | cut -c1-120
done
