#!/bin/bash
# GPU box: full -m gpu suite, smoke, default bench, and the rocprofv3 kernel summary of the same bench command.
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py 2>/dev/null | grep '^{"metric"' > gpurun_out/bench_full.json; cat gpurun_out/bench_full.json
python bench.py --pp-only --cpu-scans 0 --procs 1 --streams 4 2>/dev/null | grep '^{"metric"' > gpurun_out/bench_pp.json; cat gpurun_out/bench_pp.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_bench
# per-kernel summary of the full pipeline inside ONE process (rocprofv3 follows the rank process)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --cpu-scans 0 --procs 1 --streams 4 > gpurun_out/prof_bench.log 2>&1
ls gpurun_out/prof_bench
