#!/bin/bash
# GPU box: PP parity tests, phase timers, and the PP-only numbers with one scan at a time
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pp.py -m gpu -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MODEST_PP_DBG=128 timeout 300 python tools/pp_microbench.py 2>&1 | grep "pp3" | tail -2 | cut -c1-400
rm -rf gpurun_out/prof_pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp -o bench -- python bench.py --pp-only --cpu-scans 0 --procs 1 --streams 1 --steps 16 --warmup 2 > gpurun_out/prof_pp.log 2>&1
grep '^{"metric"' gpurun_out/prof_pp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pp-only 1 stream: %.0f scans/s, stage %.3f ms, isolated %.3f ms (frac %.4f)' % (d['value'], r['kernel_ms'], r['isolated']['kernel_ms'], r['isolated']['frac']))"
python tools/kstats.py gpurun_out/prof_pp/bench_kernel_stats.csv 6
