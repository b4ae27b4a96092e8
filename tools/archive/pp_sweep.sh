#!/bin/bash
# PP tuning sweep on the GPU box: slice records cap, stream workgroups, lane-path thresholds
cd $GRAFT_REPO_ROOT
run() { echo "$1: $(env $1 timeout 300 python tools/pp_microbench.py 2>&1 | grep ms_per_scan | tail -1 | cut -c1-60)"; }
run "X=0"
for s in 2048 3072; do run "MODEST_PP_SLICE=$s"; done
for w in 256 384 768 1024; do run "MODEST_PP_NWG=$w"; done
# dbg bits 8-15: lane groups threshold, bits 16-27: laneMax
for g in 1 3 4; do run "MODEST_PP_DBG=$((g<<8))"; done
for m in 32 128 256; do run "MODEST_PP_DBG=$((m<<16))"; done
