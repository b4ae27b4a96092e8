#!/bin/bash
# GPU box: every artefact under profiles/ in one call (copied from gpurun_out/final/ afterwards)
cd $GRAFT_REPO_ROOT
F=gpurun_out/final; rm -rf $F; mkdir -p $F
line() { grep '^{"metric"' | tail -1; }
python bench.py --cpu-scans 0 --cli-scans 0 > /dev/null 2>&1   # warm the box (clocks, page cache)
python bench.py 2>/dev/null | line > $F/bench_full.json
python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 2>/dev/null | line > $F/bench_full_1proc.json
python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 6 2>/dev/null | line > $F/bench_full_1proc_6threads.json
python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 2>/dev/null | line > $F/bench_driver_steps20.json
python bench.py --pp-only --cpu-scans 0 --cli-scans 0 2>/dev/null | line > $F/bench_pp_only.json
python bench.py --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 2>/dev/null | line > $F/bench_pp_only_1stream.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# the default command: rank process + helper processes, one output set per process, merged
rm -rf gpurun_out/prof_default
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default -o bench_%pid% -- python bench.py --cpu-scans 0 --cli-scans 0 > gpurun_out/prof_default.log 2>&1
python tools/merge_kstats.py $F/bench_full_kernel_stats.csv $(ls gpurun_out/prof_default/*kernel_stats.csv)
rm -f gpurun_out/prof_default/*kernel_trace.csv
# one scan at a time: clean per-kernel durations
rm -rf gpurun_out/prof_single
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -o bench -- python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 64 > gpurun_out/prof_single.log 2>&1
cp gpurun_out/prof_single/bench_kernel_stats.csv $F/bench_full_1proc_kernel_stats.csv
rm -rf gpurun_out/prof_pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp -o bench -- python bench.py --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 32 --warmup 4 > gpurun_out/prof_pp.log 2>&1
cp gpurun_out/prof_pp/bench_kernel_stats.csv $F/pp_only_kernel_stats.csv
rm -f gpurun_out/prof_single/*kernel_trace.csv gpurun_out/prof_pp/*kernel_trace.csv
bash tools/pp_pmc.sh > $F/pp_pmc.log 2>&1
cp gpurun_out/pp_pmc.json $F/pp_pmc_counters.json; cp gpurun_out/pp_traffic.json $F/pp_traffic.json
tail -1 $F/pp_pmc.log
for f in $F/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], '%.0f scans/s' % d['value'], 'chain of %d scans %.3f ms frac %.4f' % (r['scans_per_launch'], r['kernel_ms'], r['frac']), d.get('speedup_vs_cpu'))"; done
python tools/kstats.py $F/bench_full_1proc_kernel_stats.csv 70
# ordered launch list of one full-pipeline scan (one scan at a time)
bash tools/scan_trace.sh > /dev/null 2>&1; cp gpurun_out/scan_trace.txt $F/scan_trace.txt
# diagnostics: each half of the pipeline alone from the default process pool
python bench.py --cpu-scans 0 --cli-scans 0 --mask-only 2>/dev/null | line > $F/bench_mask_only.json
python tools/pp_frames_microbench.py 2>/dev/null | tail -1 > $F/pp_stream_microbench.json
bash tools/mask_kstats.sh > $F/mask_kernel_stats.txt 2>&1
python tools/host_profile.py 100 tottime 2>&1 | grep -v amdgpu.ids | sed "s#$GRAFT_REPO_ROOT/##g; s#/usr/local/lib/python3.10/dist-packages/##g" | head -40 | cut -c1-160 > $F/host_profile.txt
python tools/soak_mask.py 8 1500 2>&1 | grep -v amdgpu.ids | tail -3 > $F/determinism_soak.txt
