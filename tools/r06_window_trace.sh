#!/bin/bash
# round 6: the driver's 20-step window, N runs with the helpers' own phase times (MODEST_BENCH_TRACE=1): what a slow run looks like
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06_window_trace.txt
python bench.py --cpu-scans 0 --cli-scans 0 --steps 128 --sharing best > /dev/null 2>&1
for i in $(seq 1 ${N:-12}); do
  MODEST_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 --sharing best 2>gpurun_out/wt.err | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run value %.0f ms_per_step %.3f' % (d['value'], d['ms_per_step']))" >> gpurun_out/r06_window_trace.txt
  grep "^\[helper" gpurun_out/wt.err | grep " 7 steps\| 6 steps" | cut -c1-260 >> gpurun_out/r06_window_trace.txt
done
cat gpurun_out/r06_window_trace.txt
