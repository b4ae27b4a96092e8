#!/bin/bash
# parity tests of the PP stage + timing of V3 against V2 + per-kernel profile
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q -m gpu 2>&1 | tail -15
echo "== V3"; timeout 300 python tools/pp_microbench.py 2>&1 | tail -3
echo "== V1 (direct path)"; MODEST_PP_VARIANT=1 timeout 300 python tools/pp_microbench.py 2>&1 | tail -2
rm -rf gpurun_out/prof3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3 -o p -- python bench.py --steps 8 --warmup 2 --pp-only --cpu-scans 0 --streams 1 > gpurun_out/prof3.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof3/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print('%-60s calls %5s avg %9.1f us  %5s%%' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
grep '^{"metric"' gpurun_out/prof3.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pp-only 1 stream scans/s %.1f pp stage ms %.3f frac %.4f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
