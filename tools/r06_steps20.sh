#!/bin/bash
# round 6: the driver's 20-step window under different deals of the 20 scans to the helpers (MODEST_WINDOW_DEAL), three runs each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06_steps20.txt
python bench.py --cpu-scans 0 --cli-scans 0 --steps 128 --sharing best > /dev/null 2>&1
for d in ${DEALS:-"7,7,6" "9,7,4" "8,6,4,2" "7,6,4,3" "8,7,5" "7,7,6" "8,6,4,2"}; do
  for rep in 1 2 3; do
    echo "deal=$d: $(MODEST_WINDOW_DEAL=$d python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 --sharing best 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  steady %.0f  helpers %d  calls %s' % (d['value'], d['steady_state']['value'], d['config']['host_processes_per_gpu'], d['config']['pp_calls_in_timed_region']))")" >> gpurun_out/r06_steps20.txt
  done
done
cat gpurun_out/r06_steps20.txt
