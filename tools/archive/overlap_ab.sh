#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1
run() { python bench.py --cpu-scans 0 --cli-scans 0 "$@" 2>&1 | grep '^{"metric"\|Error\|error' | head -3 | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l); print('$*', round(d['value'],1), 'scans/s', d.get('parity'), d['steady_state'] and round(d['steady_state']['value']))
    except Exception: print(l[:300])"; }
for r in 1 2 3; do run --steps 600; run --steps 600 --no-prefetch; done
for r in 1 2 3 4; do run --steps 20 --warmup 5; run --steps 20 --warmup 5 --no-prefetch; done
run --steps 64 --procs 1 --streams 1; run --steps 64 --procs 1 --streams 1 --no-prefetch
