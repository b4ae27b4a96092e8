#!/bin/bash
# the driver's command (--steps 20 --warmup 5): value, spread, and the steady state of the full pool
cd $GRAFT_REPO_ROOT
for m in 2 3 4 5; do
  for rep in 1 2 3 4; do
    MODEST_MIN_STEPS_PER_HELPER=$m timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('min_steps $m rep $rep:', round(d['value']), 'scans/s, helpers', d['config']['host_processes_per_gpu'], 'ms_per_step', round(d['ms_per_step'],3), 'steady', d['steady_state'] and round(d['steady_state']['value']))"
  done
done
timeout 400 python bench.py --cpu-scans 0 --cli-scans 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('default 560 steps:', round(d['value']), 'scans/s, helpers', d['config']['host_processes_per_gpu'], 'steady', d['steady_state'])"
