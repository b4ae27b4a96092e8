#!/bin/bash
cd $GRAFT_REPO_ROOT
export MODEST_PP_FRAMES_PATH=gather-wave
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "768 256" "768 1024" "768 4096" "512 1024" "512 4096" "384 4096"; do
  set -- $cfg
  rm -rf gpurun_out/prof_sw
  MODEST_PP6_WMAX=$1 MODEST_PP6_DENSE_ITEMS=$2 MODEST_PP6_WGS=4 NSCAN=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sw -o b -- python tools/pp5_microbench.py > gpurun_out/prof_sw.log 2>&1
  echo "wmax $1 denseItems $2: $(python tools/kstats.py gpurun_out/prof_sw/b_kernel_stats.csv 2 pp6_wave_join | head -2 | cut -c1-20,60-120 | tr '\n' ' ')"
done
