#!/bin/bash
# BASELINE config 5 (nuScenes shape): bench line + per-kernel stats of the PP stage
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
python bench.py --config c5 --steps 384 --cpu-scans 1 --cpu-best-effort 0 2>gpurun_out/r04/c5.err | grep '^{"metric"' | tail -1 > gpurun_out/r04/bench_c5.json
python bench.py --config c5 --steps 384 --cpu-scans 0 --no-pp-block 2>>gpurun_out/r04/c5.err | grep '^{"metric"' | tail -1 > gpurun_out/r04/bench_c5_noblock.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o bench -- python bench.py --config c5 --pp-only --cpu-scans 0 --procs 1 --streams 1 --steps 64 --warmup 16 > gpurun_out/prof_c5.log 2>&1
cp gpurun_out/prof_c5/bench_kernel_stats.csv gpurun_out/r04/c5_pp_kernel_stats.csv; rm -rf gpurun_out/prof_c5
for f in gpurun_out/r04/bench_c5*.json; do python -c "
import json; d=json.load(open('$f')); r=d['roofline']
print('$f'.split('/')[-1], 'value %.0f ingest %.0f' % (d['value'], d['value_with_ingest']['value']), 'roofline frac %.4f ms/call %.3f scans/call %d block %s' % (r['frac'], r['kernel_ms'], r['scans_per_launch'], r['block_path']), 'parity', d.get('parity'))"; done
