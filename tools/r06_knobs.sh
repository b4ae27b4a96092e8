#!/bin/bash
# round 6: default-run `value` against the persistent-grid size per process (--pp-cus) and the helper count (--procs), alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06_knobs.txt
python bench.py --cpu-scans 0 --cli-scans 0 --steps 128 --sharing best > /dev/null 2>&1
for rep in 1 2; do
for k in "--pp-cus 128" "--pp-cus 192" "--pp-cus 0" "--pp-cus 96" "--procs 6" "--procs 7"; do
  echo "$k: $(python bench.py --cpu-scans 0 --cli-scans 0 --sharing best $k 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  value_with_ingest %.0f' % (d['value'], d['value_with_ingest']['value']))")" >> gpurun_out/r06_knobs.txt
done
done
cat gpurun_out/r06_knobs.txt
