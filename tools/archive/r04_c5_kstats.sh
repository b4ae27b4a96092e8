#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
rm -rf gpurun_out/prof_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o bench -- python bench.py --config c5 --pp-only --cpu-scans 0 --procs 1 --streams 1 --steps 64 --warmup 16 > gpurun_out/prof_c5.log 2>&1
cp gpurun_out/prof_c5/bench_kernel_stats.csv gpurun_out/r04/c5_pp_kernel_stats.csv; rm -rf gpurun_out/prof_c5
PP_BATCH=8 python tools/kstats.py gpurun_out/r04/c5_pp_kernel_stats.csv 14 ppb_join
rm -rf gpurun_out/prof_c3
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o bench -- python bench.py --no-pp-block --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 64 --warmup 16 > gpurun_out/prof_c3.log 2>&1
cp gpurun_out/prof_c3/bench_kernel_stats.csv gpurun_out/r04/pp_only_noblock_kernel_stats.csv; rm -rf gpurun_out/prof_c3
PP_BATCH=8 python tools/kstats.py gpurun_out/r04/pp_only_noblock_kernel_stats.csv 14 ppb_join
