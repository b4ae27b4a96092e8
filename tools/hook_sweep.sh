#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 0 1 2 3; do
  rm -rf gpurun_out/prof_h$r
  MODEST_HOOK_ROUNDS=$r timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h$r -o b -- python bench.py --cpu-scans 0 --procs 1 --streams 1 --steps 48 > gpurun_out/h$r.log 2>&1
  echo "rounds $r"; python tools/kstats.py gpurun_out/prof_h$r/b_kernel_stats.csv 80 | grep -E "hook_adj|flatten|union_adj|compress|label_adj|lowest|total"
done
