#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 0 1; do
  rm -rf gpurun_out/prof_k$r
  MODEST_KNN_DBG=$r timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_k$r -o b -- python bench.py --cpu-scans 0 --procs 1 --streams 1 --steps 48 > gpurun_out/k$r.log 2>&1
  echo "dbg $r"; python tools/kstats.py gpurun_out/prof_k$r/b_kernel_stats.csv 80 | grep -E "knn_kth|degree_adj|lowest|mad_k|total"
done
