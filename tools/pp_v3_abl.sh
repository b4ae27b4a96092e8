#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pp.py -x -q -m gpu 2>&1 | tail -2
for d in 0 16; do
rm -rf gpurun_out/abl
MODEST_PP_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -o a -- python tools/pp_microbench.py > gpurun_out/abl.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/abl/**/*kernel_stats.csv', recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    for k in ('pp3_stream<true>','pp3_stream<false>','pp3_join','pp3_scan','pp3_blocks','pp3_plan'):
        if k in r['Name']: out.append('%s %.1f' % (k, float(r['AverageNs'])/1e3))
print('dbg=$d', ' | '.join(sorted(out)), open('gpurun_out/abl.log').read().split('ms_per_scan')[-1][:12])
PY
done
