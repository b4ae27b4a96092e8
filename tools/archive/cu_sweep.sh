#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1
for c in 256 224 192 160 128; do
  MODEST_NUM_CUS=$c python bench.py --cpu-scans 0 --cli-scans 0 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cus $c', round(d['value'],1), 'scans/s', d['ms_per_step'])"
done
