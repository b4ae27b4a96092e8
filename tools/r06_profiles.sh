#!/bin/bash
# GPU box: the artefacts under profiles/r06_* in one call (copied from gpurun_out/r06/ afterwards).  STEPS: all | bench | kstats | pmc
cd $GRAFT_REPO_ROOT
F=gpurun_out/r06; mkdir -p $F
S=${STEPS:-all}
line() { grep '^{"metric"' | tail -1; }
if [ $S = all ] || [ $S = bench ]; then
python bench.py --cpu-scans 0 --cli-scans 0 --steps 128 --sharing best > /dev/null 2>&1   # warm the box (clocks, page cache)
python bench.py 2>$F/bench_full.err | line > $F/bench_full.json
python bench.py --steps 20 --warmup 5 --cpu-scans 0 --cli-scans 0 2>/dev/null | line > $F/bench_driver_steps20.json
python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --sharing best 2>/dev/null | line > $F/bench_full_1proc.json
python bench.py --pp-only --cpu-scans 0 --cli-scans 0 --sharing best 2>/dev/null | line > $F/bench_pp_only.json
python bench.py --mask-only --cpu-scans 0 --cli-scans 0 --sharing best 2>/dev/null | line > $F/bench_mask_only.json
python bench.py --config c5 --steps 384 --cpu-scans 1 --cpu-best-effort 0 2>/dev/null | line > $F/bench_c5.json
fi
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
if [ $S = all ] || [ $S = kstats ]; then
rm -rf gpurun_out/prof_pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pp -o bench -- python bench.py --pp-only --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 128 --warmup 16 --sharing best > gpurun_out/prof_pp.log 2>&1
cp gpurun_out/prof_pp/bench_kernel_stats.csv $F/pp_only_kernel_stats.csv; rm -rf gpurun_out/prof_pp
rm -rf gpurun_out/prof_single
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_single -o bench -- python bench.py --cpu-scans 0 --cli-scans 0 --procs 1 --streams 1 --steps 128 --sharing best > gpurun_out/prof_single.log 2>&1
cp gpurun_out/prof_single/bench_kernel_stats.csv $F/bench_full_1proc_kernel_stats.csv; rm -rf gpurun_out/prof_single
# the block path on the realistic shard (reference-rule windows, traversals entering and leaving): the kernels of `roofline`
PP_BLOCK=32 KS_NAME=r06/pp_block_realistic_kernel_stats.csv bash tools/r05_kstats.sh --matched 8,3,15 --presence 5 > $F/pp_block_realistic_kstats.txt 2>&1
fi
if [ $S = all ] || [ $S = pmc ]; then
PP_BLOCK=32 bash tools/pp4_pmc.sh > $F/pp4_pmc.log 2>&1
PP_BLOCK=32 python tools/pp4_traffic.py | tee -a $F/pp4_pmc.log
cp gpurun_out/pp4_pmc.json $F/pp_block_pmc_counters.json; cp gpurun_out/pp4_traffic.json $F/pp_block_traffic.json
export PROBE_ARGS="--matched 8,3,15 --presence 5"
python tools/pp_block_probe.py --scans 32 --reps 3 --shards 2 $PROBE_ARGS --json-out $F/probe_realistic.jsonl 2>&1 | grep 'PARITY\|^block\|sharing' > $F/probe_realistic.txt
PP_BLOCK=32 bash tools/pp4_pmc.sh > $F/pp4_pmc_realistic.log 2>&1
PP_ALG_BYTES=$(python -c "import json; print(json.loads(open('$F/probe_realistic.jsonl').readlines()[-1])['algorithmic_bytes_per_scan'])") PP_BLOCK=32 PP_TRAFFIC_OUT=$F/pp_block_traffic_realistic.json python tools/pp4_traffic.py | tee -a $F/pp4_pmc_realistic.log
cp gpurun_out/pp4_pmc.json $F/pp_block_pmc_counters_realistic.json
unset PROBE_ARGS
fi
for f in $F/bench_*.json; do python -c "
import json,sys
try:
    d=json.load(open('$f'))
except Exception as e:
    print('$f'.split('/')[-1], 'NO JSON'); sys.exit(0)
r=d['roofline']; b=d.get('roofline_best_case') or {}; c=d.get('cli') or {}
print('$f'.split('/')[-1], 'value %.0f ingest %.0f steady %s' % (d['value'], d['value_with_ingest']['value'], ('%.0f' % d['steady_state']['value']) if d.get('steady_state') else '-'), '| roofline %.4f %.3f ms / %d scans block=%s traffic %s' % (r['frac'], r['kernel_ms'], r['scans_per_launch'], r['block_path'], r.get('traffic_per_scan')), '| best case', b.get('frac'), '| path', d['config'].get('pp_path_in_timed_region'), '| cli', {k: round(v) for k,v in c.items() if k.endswith('per_s') or k.endswith('workers') and isinstance(v,(int,float))}, d.get('speedup_vs_cpu'))"; done
if [ $S = all ] || [ $S = c4 ]; then
# one rank's share of the Lyft train set (11 873 / 8 = 1 485 scans) through the CLIs, separate and fused
python bench.py --steps 64 --cpu-scans 0 --cli-scans 1485 --sharing best 2>/dev/null | line > $F/bench_c4_shard.json
python -c "
import json; d=json.load(open('$F/bench_c4_shard.json')); c=d['cli']; print('c4 shard', {k:(round(v) if isinstance(v,float) else v) for k,v in c.items() if k!='note'})"
fi
