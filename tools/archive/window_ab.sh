#!/bin/bash
# how often the driver's 20-step window is slow: N runs per mode (trial loops on the device / on the host)
cd $GRAFT_REPO_ROOT
A="--steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0"
for rep in $(seq 1 ${N:-5}); do
for h in 0 1; do
  if [ $h = 1 ]; then export MODEST_RANSAC_HOST=1; else unset MODEST_RANSAC_HOST; fi
  v=$(MODEST_BENCH_TRACE=1 timeout 250 python bench.py $A 2>gpurun_out/w.err | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.0f' % d['value'])")
  echo "host_loop=$h value $v | slowest helper: $(grep '^\[helper' gpurun_out/w.err | grep ' [23] steps' | sort -t' ' -k5 -n | tail -1 | cut -c1-100)"
done
done
