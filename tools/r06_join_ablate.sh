#!/bin/bash
# round 6: what bounds b4_join -- kernel time of the join (rocprofv3 kernel trace) for build variants and MODEST_PP4_DBG bits
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; : > gpurun_out/r06_join_ablate.txt
IFS='|' read -ra VS <<< "$VARIANTS"
for v in "${VS[@]}"; do
  export MODEST_EXTRA_CXXFLAGS="$v"
  python -c "from modest_amd import build; build.build(verbose=False)" > gpurun_out/ab_build.log 2>&1 || { echo "BUILD FAILED $v" >> gpurun_out/r06_join_ablate.txt; continue; }
  for dbg in ${DBGS:-0}; do
    rm -rf gpurun_out/prof_abl
    MODEST_PP4_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_abl -o p -- python tools/pp_block_probe.py --scans 32 --reps 3 --shards 2 $PROBE_ARGS > gpurun_out/prof_abl.log 2>&1
    f=$(find gpurun_out/prof_abl -name "*kernel_stats.csv" | head -1)
    echo "[$v] dbg=$dbg $(python -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'b4_join' in r['Name']: print('b4_join avg %.1f us' % (float(r['AverageNs'])/1e3), end=' ')
") | $(grep '^block' gpurun_out/prof_abl.log | head -1)" >> gpurun_out/r06_join_ablate.txt
  done
  unset MODEST_EXTRA_CXXFLAGS
done
rm -rf gpurun_out/prof_abl
cat gpurun_out/r06_join_ablate.txt
