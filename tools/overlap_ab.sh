#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1
run() { python bench.py --cpu-scans 0 --cli-scans 0 --steps 600 "$@" 2>&1 | grep '^{"metric"\|Error\|error' | head -3 | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l); print('$*', round(d['value'],1), 'scans/s', d.get('parity'))
    except Exception: print(l[:200])"; }
run; run --no-overlap; run; run --no-overlap
run --procs 7; run --procs 6; run --procs 1 --streams 1; run --procs 1 --streams 1 --no-overlap
