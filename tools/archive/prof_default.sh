#!/bin/bash
# GPU box: rocprofv3 kernel trace of the DEFAULT bench command (helper processes included)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -o bench_%pid% -- python bench.py --cpu-scans 0 > gpurun_out/prof_default.log 2>&1
grep '^{"metric"' gpurun_out/prof_default.log | cut -c1-220
ls gpurun_out/prof_default | head -40
python tools/merge_kstats.py gpurun_out/prof_default_kernel_stats.csv $(ls gpurun_out/prof_default/*kernel_stats.csv)
python tools/kstats.py gpurun_out/prof_default_kernel_stats.csv 12
rm -f gpurun_out/prof_default/*kernel_trace.csv
tail -5 gpurun_out/prof_default.log | cut -c1-200
