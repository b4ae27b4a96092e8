#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in 512 384 256 128; do
rm -rf gpurun_out/abl
MODEST_PP_NWG=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl -o a -- python tools/pp_microbench.py > gpurun_out/abl.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('gpurun_out/abl/**/*kernel_stats.csv', recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    for k in ('pp3_stream<true>','pp3_stream<false>','pp3_join','pp3_scan'):
        if k in r['Name']: out.append('%s %.1f' % (k, float(r['AverageNs'])/1e3))
print('nwg=$w', ' | '.join(sorted(out)), open('gpurun_out/abl.log').read().split('ms_per_scan')[-1][:12])
PY
done
