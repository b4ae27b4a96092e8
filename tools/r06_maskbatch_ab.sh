#!/bin/bash
# round 6: stages 2 + 3 in chains of 16 against 32 in the default run, alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r06_mask_batch_ab.txt
python bench.py --cpu-scans 0 --cli-scans 0 --steps 128 --sharing best > /dev/null 2>&1
for mb in 16 32 16 32 16 32; do
  echo "--mask-batch $mb: $(python bench.py --cpu-scans 0 --cli-scans 0 --sharing best --mask-batch $mb 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.0f  value_with_ingest %.0f (%d steps)' % (d['value'], d['value_with_ingest']['value'], d['value_with_ingest']['steps']))")" >> gpurun_out/r06_mask_batch_ab.txt
done
cat gpurun_out/r06_mask_batch_ab.txt
