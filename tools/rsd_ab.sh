#!/bin/bash
# A/B of the RANSAC trial loops: on the device (default) against the host loop (MODEST_RANSAC_HOST=1)
cd $GRAFT_REPO_ROOT
for rep in ${REPS:-1}; do
for h in 0 1; do
  if [ $h = 1 ]; then export MODEST_RANSAC_HOST=1; else unset MODEST_RANSAC_HOST; fi
  echo "== host loop $h"
  tools/bench_quick.sh "--steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0" "--steps 560 --warmup 16 --cpu-scans 0 --cli-scans 0" "--procs 1 --streams 1 --steps 256 --warmup 16 --cpu-scans 0 --cli-scans 0" 2>&1 | grep -v "^W2026\|^E2026"
done
done
