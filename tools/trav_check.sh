#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pp.py -m gpu -q -x 2>&1 | tail -2
for cfg in "--traversals 40 --frames 6" "--traversals 64 --frames 4" "--traversals 10 --frames 36"; do
  python bench.py $cfg --pp-only --steps 32 --warmup 4 --procs 1 --streams 1 --cpu-scans 1 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$cfg ->', '%.0f scans/s' % d['value'], 'stage %.3f ms' % r['kernel_ms'], 'parity', d['parity'])"
done
