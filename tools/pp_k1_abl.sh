#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 64 128 192; do
rm -rf gpurun_out/abl
MODEST_PP_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -o a -- python bench.py --steps 6 --warmup 2 --pp-only --cpu-scans 0 > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/abl/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'pp2_route' in r['Name']: print('dbg=$d pp2_route %.1f us' % (float(r['AverageNs'])/1e3))
PY
done
