#!/bin/bash
# the driver's 20-step window N times with allocation tracing: prints the slow runs' helper lines and any allocation inside the window
cd $GRAFT_REPO_ROOT
A="--steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0"
for rep in $(seq 1 ${N:-8}); do
  v=$(MODEST_ALLOC_TRACE=1 MODEST_BENCH_TRACE=1 timeout 250 python bench.py $A 2>gpurun_out/w.err | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.0f' % d['value'])")
  echo "run $rep value $v"
  grep -n "^\[helper.* [23] steps\|modest alloc" gpurun_out/w.err | awk -F: '{print $1": "$2$3$4}' | cut -c1-160 | tail -14
done
