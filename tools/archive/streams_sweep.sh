#!/bin/bash
# full-pipeline throughput vs scans in flight per GPU
mkdir -p gpurun_out
for s in 2 4 6 8 12; do
  python bench.py --steps 48 --warmup 6 --cpu-scans 0 --streams $s 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', $s, 'scans/s %.1f' % d['value'], 'pp ms %.3f' % d['roofline']['kernel_ms'])"
done | tee gpurun_out/streams_sweep.txt
