#!/bin/bash
# GPU box: PP parity tests + the PP-only bench line (isolated stage time) + clean single-stream profile
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pp.py -m gpu -q -x 2>&1 | tail -2
python bench.py --pp-only --cpu-scans 0 --procs 1 --streams 1 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pp-only 1 stream: %.0f scans/s, stage %.3f ms, isolated %.3f ms (frac %.4f)' % (d['value'], r['kernel_ms'], r['isolated']['kernel_ms'], r['isolated']['frac']))"
bash tools/prof_single.sh 2>&1 | grep -E "passed|failed|pp3_|pp_|total"
