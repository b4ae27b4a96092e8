#!/bin/bash
# GPU box: frame-store PP parity tests, then timings of the two frame paths
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frames.py -m gpu -q -x 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for path in stream gather-wave; do
  echo "== path $path"
  if [ $path = stream ]; then unset MODEST_PP_FRAMES_PATH; else export MODEST_PP_FRAMES_PATH=$path; fi
  timeout 300 python tools/pp5_microbench.py 2>&1 | tail -3 | cut -c1-200
done
export MODEST_PP_FRAMES_PATH=gather-wave
rm -rf gpurun_out/prof_pp5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp5 -o b -- python tools/pp5_microbench.py > gpurun_out/prof_pp5.log 2>&1
python tools/kstats.py gpurun_out/prof_pp5/b_kernel_stats.csv 10 pp6_wave_join
