#!/bin/bash
# GPU box: PP parity tests, then a rocprofv3 kernel-trace of the PP-only bench.
timeout 600 python -m pytest tests/test_gpu_pp.py -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp -o pp -- python bench.py --steps 10 --warmup 2 --pp-only --cpu-scans 0 > gpurun_out/prof_pp.log 2>&1
grep "^{\"metric" gpurun_out/prof_pp.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])"
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_pp/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:58].ljust(58), r['Calls'].rjust(5), ('%.1f us' % (float(r['AverageNs'])/1e3)).rjust(12), r['Percentage'])
PY
