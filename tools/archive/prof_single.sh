#!/bin/bash
# GPU box: stage tests, then the rocprofv3 kernel summary of the full pipeline with ONE scan in flight
# (clean per-kernel durations: nothing else shares the GPU).
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_single
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -o bench -- python bench.py --cpu-scans 0 --procs 1 --streams 1 --steps 64 > gpurun_out/prof_single.log 2>&1
grep '^{"metric"' gpurun_out/prof_single.log | cut -c1-200
python tools/kstats.py gpurun_out/prof_single/bench_kernel_stats.csv 60
