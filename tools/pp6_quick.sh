#!/bin/bash
cd $GRAFT_REPO_ROOT
export MODEST_PP_FRAMES_PATH=gather-wave
timeout 300 python -m pytest tests/test_gpu_frames.py -m gpu -q -x -k "gather-wave" 2>&1 | tail -2
NSCAN=1 MODEST_PP5_PROF=1 timeout 120 python tools/pp5_microbench.py 2>&1 | grep "^\[pp\|equal" | head -4
NSCAN=3 timeout 120 python tools/pp5_microbench.py 2>&1 | tail -2 | cut -c1-120
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pp6
NSCAN=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp6 -o b -- python tools/pp5_microbench.py > gpurun_out/prof_pp6.log 2>&1
python tools/kstats.py gpurun_out/prof_pp6/b_kernel_stats.csv 5 pp6_wave_join
