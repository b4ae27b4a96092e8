#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1
run() { python bench.py --cpu-scans 0 --cli-scans 0 --steps 600 "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), 'scans/s')"; }
for c in 48 64 96 128 192; do
  for p in 7 8; do run --procs $p --pp-cus $c; done
done
run --procs 8 --pp-cus 64 --mask-only
run --procs 8 --pp-cus 64 --pp-only
run --procs 4 --streams 2 --pp-cus 64
