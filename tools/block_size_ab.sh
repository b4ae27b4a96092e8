#!/bin/bash
# default pipeline run with blocks of 32 against blocks of 16, alternating on ONE box
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for cfg in "--pp-batch 32 --shard-scans 32 --scans 64" "--pp-batch 16 --shard-scans 16 --scans 32"; do
    python bench.py --cpu-scans 0 --cli-scans 0 $cfg 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$cfg', 'value', round(d['value']), 'ingest', round(d['value_with_ingest']['value']), 'roof', round(r['frac'],4))"
  done
done
