#!/bin/bash
# GPU box: per-kernel GPU time of stages 2 + 3 of one scan (one process, one stream, nothing else on the GPU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_mask
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mask -o m -- python bench.py --mask-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 128 --warmup 8 > gpurun_out/prof_mask.log 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_mask/m_kernel_stats.csv')))
import os
B = int(os.environ.get('PP_BATCH', '4'))   # scans per chain (bench.py --pp-batch)
ns = sum(int(r['Calls']) for r in rows if 'mask_count_kernel' in r['Name']) + B * sum(int(r['Calls']) for r in rows if 'mcb_mask_count' in r['Name'])
tot, nl = 0.0, 0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    if any(k in r['Name'] for k in ('ppb_', 'pp3_', 'pp_', 'frame_sort')):
        continue
    per = float(r['TotalDurationNs']) / ns / 1e3
    tot += per
    nl += int(r['Calls'])
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:48]:48s} calls/scan {int(r['Calls'])/ns:4.1f} avg {float(r['AverageNs'])/1e3:6.1f} us  per-scan {per:6.1f}")
print(f"mask + box + label stages: {tot:.1f} us of GPU time per scan in {nl/ns:.1f} launches ({ns} scans)")
PY
