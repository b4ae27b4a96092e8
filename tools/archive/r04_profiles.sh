#!/bin/bash
# GPU box: the artefacts under profiles/r04_* in one call (copied from gpurun_out/r04/ afterwards)
cd $GRAFT_REPO_ROOT
F=gpurun_out/r04; mkdir -p $F
line() { grep '^{"metric"' | tail -1; }
python bench.py --cpu-scans 0 --cli-scans 0 --steps 128 > /dev/null 2>&1   # warm the box (clocks, page cache)
python bench.py 2>$F/bench_full.err | line > $F/bench_full.json
python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 2>/dev/null | line > $F/bench_driver_steps20.json
python bench.py --cpu-scans 0 --cli-scans 0 --no-pp-block 2>/dev/null | line > $F/bench_full_noblock.json
python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 2>/dev/null | line > $F/bench_full_1proc.json
python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 6 2>/dev/null | line > $F/bench_full_1proc_6threads.json
python bench.py --pp-only --cpu-scans 0 --cli-scans 0 2>/dev/null | line > $F/bench_pp_only.json
python bench.py --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 2>/dev/null | line > $F/bench_pp_only_1stream.json
python bench.py --mask-only --cpu-scans 0 --cli-scans 0 2>/dev/null | line > $F/bench_mask_only.json
# one rank's share of the 11 873-scan Lyft train set (C4: 11 873 / 8 = 1 485 scans), three CLIs, workers=8
python bench.py --steps 64 --cpu-scans 0 --cli-scans 1485 2>$F/bench_c4_shard.err | line > $F/bench_c4_shard.json
python bench.py --config c5 --steps 384 --cpu-scans 1 --cpu-best-effort 0 2>/dev/null | line > $F/bench_c5.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp -o bench -- python bench.py --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 128 --warmup 16 > gpurun_out/prof_pp.log 2>&1
cp gpurun_out/prof_pp/bench_kernel_stats.csv $F/pp_only_kernel_stats.csv; rm -rf gpurun_out/prof_pp
rm -rf gpurun_out/prof_single
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -o bench -- python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 128 > gpurun_out/prof_single.log 2>&1
cp gpurun_out/prof_single/bench_kernel_stats.csv $F/bench_full_1proc_kernel_stats.csv; rm -rf gpurun_out/prof_single
PP_BLOCK=16 bash tools/pp4_pmc.sh > $F/pp4_pmc.log 2>&1
PP_BLOCK=16 python tools/pp4_traffic.py | tee -a $F/pp4_pmc.log
cp gpurun_out/pp4_pmc.json $F/pp_block_pmc_counters.json; cp gpurun_out/pp4_traffic.json $F/pp_block_traffic.json
MODEST_PP4_DBG=512 python tools/pp_block_probe.py --scans 16 --reps 1 --shards 1 2>&1 | grep "b4_join" | head -1 > $F/pp_block_join_phases.txt
for f in $F/bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f'))
except Exception as e:
    print('$f'.split('/')[-1], 'NO JSON'); sys.exit(0)
r=d['roofline']; c=d.get('cli') or {}
print('$f'.split('/')[-1], 'value %.0f ingest %.0f steady %s' % (d['value'], d['value_with_ingest']['value'], ('%.0f' % d['steady_state']['value']) if d.get('steady_state') else '-'), '| roofline %.4f %.3f ms / %d scans block=%s' % (r['frac'], r['kernel_ms'], r['scans_per_launch'], r['block_path']), '| cli', {k: round(v) for k,v in c.items() if k.endswith('per_s') or k.endswith('workers') and isinstance(v,(int,float))}, d.get('speedup_vs_cpu'))"; done
