#!/bin/bash
# the driver's 20-step window under different helper deals (MODEST_MIN_STEPS_PER_HELPER k -> min(8, 20 // k) helpers):
# k=2: 8 x 2-3 scans (per-scan chain), k=7 and 10: 2 x 10 (block path), k=20: 1 x 20 (block of 16 + chain of 4)
cd $GRAFT_REPO_ROOT
for k in ${KS:-2 10 20 5}; do
  for r in 1 2 3; do
    MODEST_MIN_STEPS_PER_HELPER=$k python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('k=$k run $r value %.0f steady %s ingest %.0f path %s calls %s helpers %d' % (d['value'], ('%.0f' % d['steady_state']['value']) if d.get('steady_state') else '-', d['value_with_ingest']['value'], d['config']['pp_path_in_timed_region'], d['config']['pp_calls_in_timed_region']['scans_per_call'], d['config']['host_processes_per_gpu']))"
  done
done
