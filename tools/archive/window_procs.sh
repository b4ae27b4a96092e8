#!/bin/bash
# the driver's 20-step window, N runs per helper count: value and the slowest helper's time
cd $GRAFT_REPO_ROOT
for p in ${PROCS:-8 7 6}; do
for rep in $(seq 1 ${N:-8}); do
  v=$(MODEST_BENCH_TRACE=1 timeout 250 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 --procs $p ${EXTRA:-} 2>gpurun_out/w.err | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.0f steady %.0f' % (d['value'], (d.get('steady_state') or {}).get('value', 0)))")
  echo "procs $p value $v | $(grep '^\[helper' gpurun_out/w.err | grep -v ' 24 steps\| 4[0-9] steps\| 3[0-9] steps' | awk '{print $5}' | sort -n | tr '\n' ' ')"
done
done
