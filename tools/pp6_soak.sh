#!/bin/bash
# GPU box: wave-path parity, then repeated processes of the microbench (fault soak) and timing
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_frames.py -m gpu -q -x 2>&1 | tail -3
export MODEST_PP_FRAMES_PATH=gather-wave
bad=0
for i in $(seq 1 ${1:-14}); do
  timeout 200 python tools/pp5_microbench.py > /tmp/soak.log 2>&1 || { bad=$((bad+1)); tail -2 /tmp/soak.log | cut -c1-200; }
done
echo "soak: $bad failures"
tail -3 /tmp/soak.log | cut -c1-220
unset MODEST_PP_FRAMES_PATH
timeout 200 python tools/pp5_microbench.py 2>&1 | tail -2 | cut -c1-220
