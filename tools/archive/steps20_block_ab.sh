#!/bin/bash
# the driver's 20-step window with shards / blocks of 32 against 16, alternating on ONE box
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for cfg in "--pp-batch 32 --shard-scans 32 --scans 64" "--pp-batch 16 --shard-scans 16 --scans 32"; do
    python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 $cfg 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', 'value', round(d['value']), 'steady', round(d['steady_state']['value']), 'ingest', round(d['value_with_ingest']['value']), d['config']['pp_calls_in_timed_region']['scans_per_call'], d['config']['startup']['pool_seconds'])"
  done
done
