#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1
for args in "--mask-only" "--mask-only --procs 1 --streams 1" "--mask-only --procs 12" "--mask-only --procs 16" "" "--pp-only"; do
  python bench.py --cpu-scans 0 --cli-scans 0 $args 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args', round(d['value'],1), 'scans/s', round(1e3/d['value'],1), 'us/scan', d['config'].get('host_processes'), d['config'].get('streams_per_process'))"
done
