#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1
run() { python bench.py --cpu-scans 0 --cli-scans 0 --steps 600 "$@" 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cus=$MODEST_NUM_CUS $*', round(d['value'],1), 'scans/s')"; }
for c in 128 96 64 48 32; do
  export MODEST_NUM_CUS=$c
  run --procs 8; run --procs 8
done
export MODEST_NUM_CUS=96
run --procs 8 --pp-only
export MODEST_NUM_CUS=256
run --procs 8 --pp-only
