#!/bin/bash
# GPU box: the full pipeline at other scan shapes (KITTI-size live scans, few / many traversals, nuScenes-like sparse scans)
cd $GRAFT_REPO_ROOT
for cfg in "--n-live 120000 --traversals 5 --frames 20" "--n-live 30000 --traversals 32 --frames 8" "--n-live 30000 --traversals 40 --frames 6" "--n-live 8000 --traversals 3 --frames 10" "--n-live 60000 --traversals 2 --frames 60"; do
  python bench.py $cfg --steps 24 --warmup 4 --procs 2 --scans 2 --cpu-scans 1 --cli-scans 0 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg ->', '%.0f scans/s' % d['value'], 'parity', d['parity'], 'note', d['config'].get('note'))"
done
