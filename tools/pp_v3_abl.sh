#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 1 2 4; do
rm -rf gpurun_out/abl
MODEST_PP_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -o a -- python tools/pp_microbench.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/abl/**/*kernel_stats.csv', recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    for k in ('pp3_stream<true>','pp3_stream<false>','pp3_join'):
        if k in r['Name']: out.append('%s avg %.1f min %.1f max %.1f' % (k, float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
print('dbg=$d', ' | '.join(out))
PY
done
