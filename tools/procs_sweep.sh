#!/bin/bash
# throughput and in-region PP stage time vs host processes x threads per GPU
for cfg in "1 4" "2 2" "4 1" "6 1" "7 1" "8 1" "9 1"; do
  set -- $cfg
  python bench.py --procs $1 --streams $2 --steps 240 --cpu-scans 0 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('procs $1 threads $2: %.0f scans/s, PP stage in region %.3f ms (frac %.3f), isolated %.3f ms' % (d['value'], r['kernel_ms'], r['frac'], r['isolated']['kernel_ms']))"
done
